#!/bin/bash
# round 6, call s: the SQ counter passes of round 5's miner (variant library) for the comparison with call r
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export CDML_LIB_PATH=$ROOT/build/variants/libcdml_r5mine.so
timeout -k 10 400 bash tools/pmc_passes.sh r06r_miner_r5 tools/mine_probe.py > $O/r06r_pmc_r5.log 2>&1
echo "[r06s] pmc r5 rc=$?"; tail -3 $O/r06r_miner_r5/pass0.err | cut -c1-200
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06r_miner_r5_counters.csv')))
for r in rows:
    if '11,' in r['kernel']:
        print('r5', r['kernel'][:60], 'n=%s avg_us=%s' % (r['dispatches'], r['avg_us']), {k: r[k] for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_MFMA','SQ_INSTS_VMEM','mfma_busy','clock_GHz') if k in r})
PY
