#!/bin/bash
# round 5, GPU call M: the exchange's un-permute pass writing planes (split-fp32 path): whole suite, then the N > 1 step form on one GPU
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/r05m_gpu_tests.txt 2>&1
echo "[r05m] gpu suite rc=$? $(tail -1 $O/r05m_gpu_tests.txt)"; grep "^FAILED" $O/r05m_gpu_tests.txt | head
python bench.py --steps 20 --warmup 5 --extras dp_form_one_gpu --no-cpu-baseline > $O/r05m_bench.json 2> $O/r05m.err
python -c "
import json; d=json.load(open('$O/r05m_bench.json')); print('headline', d['ms_per_step']); print({k:(v['ms_per_step'] if isinstance(v,dict) else v) for k,v in d['dp_form_one_gpu'].items() if k in ('bucketed','two','single','error')})"
