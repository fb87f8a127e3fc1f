#!/bin/bash
# round 5, GPU call B: the fused semi-hard miner (tests, config-2 step + kernel statistics), staggered half tiles A/B,
# gather sweeps (steps per launch x store policy x id span).
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "semihard" > $O/r05b_tests.txt 2>&1
echo "[r05b] semihard tests rc=$? $(tail -1 $O/r05b_tests.txt)"
python -m pytest tests/test_gpu_fullsize.py -x -q -k "config2" >> $O/r05b_tests.txt 2>&1
echo "[r05b] config2 fullsize rc=$? $(tail -1 $O/r05b_tests.txt)"
python bench.py --mode semihard --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/r05b_config2_bench.json 2> $O/r05b_c2.err
echo "[r05b] config2 bench rc=$?"; python -c "
import json; d=json.load(open('$O/r05b_config2_bench.json')); print(d['ms_per_step'], d['value'], d['kernels'])"
CDML_MINE_FUSED=0 python bench.py --mode semihard --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/r05b_config2_bench_unfused.json 2>> $O/r05b_c2.err
python -c "
import json; d=json.load(open('$O/r05b_config2_bench_unfused.json')); print('unfused', d['ms_per_step'], d['value'], d['kernels'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05b_c2 -o stats -- python3 $ROOT/bench.py --mode semihard --steps 30 --warmup 5 --no-settle --no-cpu-baseline --no-extras > $O/r05b_config2_bench_under_rocprof.json 2>> $O/r05b_c2.err
cp $(find $O/r05b_c2 -name '*kernel_stats.csv' | head -1) $O/r05b_config2_kernel_stats.csv; rm -rf $O/r05b_c2
echo "[r05b] config2 stats done"; head -8 $O/r05b_config2_kernel_stats.csv | cut -c1-200
cd $ROOT
for i in 1 2; do for S in 0 1; do
  CDML_X3_STAGGER=$S python bench.py --rows 1000000 --batch 4096 --steps 200 --warmup 10 --no-extras --no-cpu-baseline > $O/r05b_stagger_${S}_$i.json 2>> $O/r05b_st.err
  python -c "
import json; d=json.load(open('$O/r05b_stagger_${S}_$i.json')); print('stagger=$S run $i', d['ms_per_step'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','dH1_ms','dW1_ms')})" | tee -a $O/r05b_stagger.txt
done; done
python tools/gather_sweep.py --kind x3 --steps 1,2,3,4,8 > $O/r05b_gather_sweep.txt 2>&1
CDML_LIB_PATH=build/variants/libcdml_gather_ntstore.so python tools/gather_sweep.py --kind x3 --steps 1,2,3,4,8 >> $O/r05b_gather_sweep.txt 2>&1
python tools/gather_sweep.py --kind x3 --steps 2 --spans 0.01,0.1,0.3,1 >> $O/r05b_gather_sweep.txt 2>&1
python tools/gather_sweep.py --kind f16 --mode 0 --steps 1,2,3,4,8 >> $O/r05b_gather_sweep.txt 2>&1
CDML_LIB_PATH=build/variants/libcdml_gather_ntstore.so python tools/gather_sweep.py --kind f16 --mode 0 --steps 1,2,3,4,8 >> $O/r05b_gather_sweep.txt 2>&1
python tools/gather_sweep.py --kind f16 --mode 1 --steps 2 --spans 0.01,0.1,0.3,1 >> $O/r05b_gather_sweep.txt 2>&1
echo "[r05b] sweeps done"; cat $O/r05b_gather_sweep.txt | cut -c1-190
