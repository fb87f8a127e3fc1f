#!/bin/bash
# Build here (the prebuilt .so travels with the snapshot and must match the sources: cdml_build_id), then run on the GPU box.
# usage: tools/gpu.sh TIMEOUT_S 'command' [tag]
# Rule (VERDICT r5 #8 -- the round-5 RCCL-watchdog abort left no trace): a call that does not end with rc 0 keeps its
# record.  What gpurun printed, the verdict file and every log under gpurun_out/ that the call touched and that names a
# fault (abort, HIP error, watchdog, a failed test) are copied to profiles/faults/<UTC time>_<tag>/ BEFORE the next call
# can overwrite them.  Nothing is re-run to "see it again".
cd "$(dirname "$0")/.."
TAG=${3:-call}
mkdir -p gpurun_out
python __graft_entry__.py | tail -1
START=$(date +%s)
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1 | tee gpurun_out/${TAG}.gpurun.log
RC=${PIPESTATUS[0]}
# (a run script of tools/runs/ reports its steps as "[tag] step rc=N": a step that failed inside a call that itself ended
#  with rc 0 -- a failed test file, a probe that raised -- keeps its record too)
STEP_FAILED=0
grep -q -E "^\[[a-z0-9_]+\] .*rc=[1-9]" gpurun_out/${TAG}.gpurun.log && STEP_FAILED=1
if { [ "$RC" -ne 0 ] && [ "$RC" -ne 2 ] && [ "$RC" -ne 3 ]; } || [ "$STEP_FAILED" -eq 1 ]; then      # (2 = refused, 3 = no box: nothing ran)
  D=profiles/faults/$(date -u +%Y%m%dT%H%M%SZ)_${TAG}
  mkdir -p "$D"
  cp gpurun_out/${TAG}.gpurun.log "$D/" 2>/dev/null
  cp gpurun_out/.last_call.json "$D/last_call.json" 2>/dev/null
  echo "$2" > "$D/command.txt"
  git rev-parse HEAD > "$D/commit.txt" 2>/dev/null
  # logs of this call (newer than its start) that name a fault, cut to their last 400 lines
  find gpurun_out -maxdepth 1 -type f -newermt "@$START" \( -name '*.log' -o -name '*.txt' -o -name '*.err' \) | while read -r f; do
    if grep -q -i -E "abort|hipError|watchdog|CapturedEvent|Segmentation|core dumped|FAILED|Traceback|Memory access fault" "$f"; then
      tail -400 "$f" > "$D/$(basename "$f")"
    fi
  done
  echo "[gpu.sh] rc=$RC step_failed=$STEP_FAILED: record kept under $D"
fi
exit $RC
