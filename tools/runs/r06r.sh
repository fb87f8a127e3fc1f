#!/bin/bash
# round 6, call r: smoke(); SQ counter passes of the miner (round 5's kernel against the final one): instructions per launch
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python __graft_entry__.py --smoke > $O/r06r_smoke.txt 2>&1
echo "[r06r] smoke rc=$? $(grep -c 'smoke\] .* ok' $O/r06r_smoke.txt) paths ok"; tail -2 $O/r06r_smoke.txt
export CDML_LIB_PATH=$ROOT/build/variants/libcdml_r5mine.so
timeout -k 10 400 bash tools/pmc_passes.sh r06r_miner_r5 tools/mine_probe.py > $O/r06r_pmc_r5.log 2>&1
echo "[r06r] pmc r5 rc=$?"
unset CDML_LIB_PATH
timeout -k 10 400 bash tools/pmc_passes.sh r06r_miner_r6 tools/mine_probe.py > $O/r06r_pmc_r6.log 2>&1
echo "[r06r] pmc r6 rc=$?"
python3 - <<'PY'
import csv
for tag in ('r5','r6'):
    rows=list(csv.DictReader(open('gpurun_out/r06r_miner_%s_counters.csv'%tag)))
    for r in rows:
        if '11,' in r['kernel'] or 'mine' in r['kernel'] or 'semihard' in r['kernel']:
            print(tag, r['kernel'][:60], 'n=%s avg_us=%s' % (r['dispatches'], r['avg_us']), {k: r[k] for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_MFMA','SQ_INSTS_VMEM','mfma_busy','clock_GHz') if k in r})
PY
