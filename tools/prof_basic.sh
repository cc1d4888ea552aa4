cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_basic -- python bench.py --gpus 1 --steps 3 --warmup 1 --task basic --n-envs 8 --hidden 256 --no-extras --no-cpu-baseline > gpurun_out/r03_basic.log 2>&1
grep -E "^\{" gpurun_out/r03_basic.log | cut -c1-600
cp $(ls -t gpurun_out/r03_basic/*/*kernel_stats.csv | head -1) gpurun_out/r03_basic_kernel_stats.csv
head -12 gpurun_out/r03_basic_kernel_stats.csv | cut -c1-260
