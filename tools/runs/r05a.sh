#!/bin/bash
# round 5, GPU call A: baseline of the restructured bench line, graph-vs-eager A/B + traces, fp16 gather variants,
# config-2 kernel statistics.  usage (GPU box, repo root): bash tools/runs/r05a.sh
set -o pipefail
ROOT=$(pwd)
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_abi_host.py "tests/test_gpu_dist.py::test_two_rank_trainer_saves_once" \
  "tests/test_gpu_dist.py::test_overflow_on_one_rank_raises_on_every_rank_and_nothing_is_saved" -x -q > $O/r05a_tests.txt 2>&1
echo "[r05a] tests rc=$? $(tail -1 $O/r05a_tests.txt)"
python bench.py --steps 20 --warmup 5 > $O/r05a_bench.json 2> $O/r05a_bench.err
echo "[r05a] bench rc=$? $(wc -c < $O/r05a_bench.json) bytes"
python tools/graph_vs_eager.py --blocks 4 --steps 200 > $O/r05a_graph_vs_eager.txt 2>&1
echo "[r05a] graph_vs_eager rc=$?"; cat $O/r05a_graph_vs_eager.txt | tail -6
cd /tmp
for W in x3 bf16; do for FORM in eager graph; do
  rocprofv3 --kernel-trace --output-format csv -d $O/r05a_trace_${W}_$FORM -o t -- python3 $ROOT/tools/graph_vs_eager.py --trace $FORM --workload $W > $O/r05a_trace_${W}_$FORM.out 2>&1
  f=$(find $O/r05a_trace_${W}_$FORM -name '*kernel_trace.csv' | head -1)
  python3 $ROOT/tools/graph_vs_eager.py --gaps $f >> $O/r05a_graph_vs_eager_gaps.txt 2>&1
  rm -rf $O/r05a_trace_${W}_$FORM
  echo "[r05a] trace $W $FORM done"
done; done
cd $ROOT
python tools/gather_f16_bench.py 10 > $O/r05a_gather_f16.txt 2>&1
for T in gather_f16_rpw2 gather_f16_rpw8 gather_grid4 gather_grid16 gather_ntstore; do
  CDML_LIB_PATH=build/variants/libcdml_$T.so python tools/gather_f16_bench.py 10 >> $O/r05a_gather_f16.txt 2>&1
done
echo "[r05a] gather variants done"; grep "mode=0 steps/launch=4" $O/r05a_gather_f16.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05a_c2 -o stats -- python3 $ROOT/bench.py --mode semihard --steps 30 --warmup 5 --no-settle --no-cpu-baseline --no-extras > $O/r05a_config2_bench_under_rocprof.json 2> $O/r05a_c2.err
cp $(find $O/r05a_c2 -name '*kernel_stats.csv' | head -1) $O/r05a_config2_kernel_stats.csv; rm -rf $O/r05a_c2
echo "[r05a] config2 stats done"; head -12 $O/r05a_config2_kernel_stats.csv
