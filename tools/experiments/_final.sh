cd /root/repo
timeout -k 10 700 python -m pytest tests -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; tail -3 gpurun_out/final_gpu_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/final_default.json 2> gpurun_out/final_default.err; echo "bench rc $?"
timeout -k 10 200 bash tools/profile_round.sh r03x3b --precision f32x3 > gpurun_out/r03x3b_profile.log 2>&1; echo "profile rc $?"
timeout -k 10 120 python bench.py --mode predict --precision f32x3 > gpurun_out/x3_predict.json 2> gpurun_out/x3_predict.err; echo "predict rc $?"
