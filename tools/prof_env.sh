cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "full_size or many_envs" 2>&1 | tail -3
python tools/env_sweep.py --tasks gridworld,ball3d --sizes 4096,4194304 --iters 3 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_env -- python tools/env_sweep.py --tasks gridworld,push,ball3d --sizes 4194304 --iters 2 > gpurun_out/prof_env.log 2>&1
find gpurun_out/prof_env -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220 | head -30
