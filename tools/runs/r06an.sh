#!/bin/bash
# round 6, call an: the Trainer policy test (learns, evaluates, checkpoints, resumes, early stop) on both plane forms
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "trainer_learns" > $O/r06an_tests.txt 2>&1
echo "[r06an] tests rc=$? $(tail -1 $O/r06an_tests.txt)"; (grep -E "^(FAILED|ERROR)|^E  " $O/r06an_tests.txt | cut -c1-500 | head -14) || true
