#!/bin/bash
# round 6, call aj: f16x2 under use_graph (check steps eager, re-capture on a moved scale): replay == eager; the parity file on the final Python
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "graph_replay or config0 or semihard" > $O/r06aj_tests.txt 2>&1
echo "[r06aj] tests rc=$? $(tail -1 $O/r06aj_tests.txt)"; (grep -E "^(FAILED|ERROR)|^E  " $O/r06aj_tests.txt | cut -c1-400 | head -12) || true
timeout -k 10 400 python bench.py --precision f16x2 --graph --steps 100 --warmup 10 > $O/r06aj_graph.json 2> $O/r06aj_graph.err
echo "[r06aj] graph bench rc=$?"; tail -2 $O/r06aj_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r06aj_graph.json')); print(d['value'], d['ms_per_step'], d['config']['hipgraph'])"
