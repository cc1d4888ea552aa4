mkdir -p gpurun_out
( timeout 900 python tools/threshold_runs.py --tasks gridworld --schedules literal --seeds 1,2,3,4,5 --out gpurun_out/r06_thr_gridworld_literal.json 2>&1 | grep -v amdgpu.ids | cut -c1-400 ) > gpurun_out/r06_thr.log
( TMA_NO_PERSIST=1 timeout 900 python tools/threshold_runs.py --tasks gridworld --schedules literal --seeds 1,2,3 --out gpurun_out/r06_thr_gridworld_literal_launches.json 2>&1 | grep -v amdgpu.ids | cut -c1-400 ) >> gpurun_out/r06_thr.log
( timeout 2400 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r06_dist.log
cat gpurun_out/r06_thr.log gpurun_out/r06_dist.log
