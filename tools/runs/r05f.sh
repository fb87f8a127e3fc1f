#!/bin/bash
# round 5, GPU call F: the whole -m gpu suite, the miner probe, config 2 and the headline (re-run after the bit_cast fix)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/r05f_gpu_tests.txt 2>&1
echo "[r05f] gpu suite rc=$? $(tail -1 $O/r05f_gpu_tests.txt)"; grep "^FAILED" $O/r05f_gpu_tests.txt | head
python tools/mine_probe.py 2>&1 | grep "B=" | tee $O/r05f_mine_probe.txt
python bench.py --mode semihard --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/r05f_config2_bench.json 2> $O/r05f.err
python -c "
import json; d=json.load(open('$O/r05f_config2_bench.json')); print('config2', d['ms_per_step'], d['value'], d['kernels'])"
python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $O/r05f_headline.json 2>> $O/r05f.err
python -c "
import json; d=json.load(open('$O/r05f_headline.json')); print('headline', d['ms_per_step'], d['value'], d['kernels'], d['gather']['frac'])"
