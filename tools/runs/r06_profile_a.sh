#!/bin/bash
# round 6, the evidence call (part a: the headline and config 1): rocprofv3 kernel statistics + FETCH_SIZE / WRITE_SIZE + MFMA busy / clock of the final kernels
# for the headline (10 M rows, B = 8192, f32x3), config 1 (f32x3 and fp32 MFMA) and config 4 (bf16), each stamped with the
# csrc hash and its workload.  usage (GPU box, repo root): bash tools/runs/r06_profile.sh COMMIT
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
C=${1:-HEAD}
D="2026-10-05 (round 6)"
run() { # tag pmc_name workload args...
  local tag=$1 name=$2 wl=$3; shift 3
  bash tools/profile_round.sh $tag "$@" > $O/${tag}_profile.log 2>&1
  echo "[profile] $tag rc=$?"; tail -2 $O/${tag}_profile.log
  cp $O/${tag}_pmc_fetch_write.csv $O/$name.csv
  python3 tools/pmc_stamp.py $name "$D" $C "tools/profile_round.sh $tag $*" "$wl"
  rm -rf $O/$tag
}
run r06_headline latest_pmc_x3 "rows=10000000 batch=8192 mode=inbatch" --steps 100 --warmup 10
run r06_config1 latest_pmc_x3_config1 "rows=1000000 batch=4096 mode=inbatch" --rows 1000000 --batch 4096 --steps 200 --warmup 10
ls $O | grep r06_ | head -50
