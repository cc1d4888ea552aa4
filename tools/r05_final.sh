#!/bin/bash
# round 5, final measurement pass on ONE box: the driver's default bench command (line + extras), the round's rocprof / PMC summaries
# (tools/prof_round.sh), the three-term split kernel's and the literal-batch 256x256 path's kernel statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
python bench.py > gpurun_out/r05f_bench_line.json 2> gpurun_out/r05f_bench.err
echo "default bench: $(( $(date +%s) - t0 )) s, line $(wc -c < gpurun_out/r05f_bench_line.json) bytes"
cp bench_extras.json gpurun_out/r05f_bench_extras.json 2>/dev/null
bash tools/prof_round.sh r05 > gpurun_out/r05f_prof_round.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_bench_gridworld_bf16x3 -- python bench.py --gpus 1 --steps 2 --warmup 1 --hidden 256 --mfma-dtype bf16x3 --no-extras --no-cpu-baseline > gpurun_out/r05_bench_gridworld_bf16x3.log 2>&1
cp $(ls -t gpurun_out/r05_bench_gridworld_bf16x3/*/*kernel_stats.csv | head -1) gpurun_out/r05_bench_gridworld_bf16x3_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_literal256_h256 -- python tools/time_literal256.py gridworld 4096 256 256 > gpurun_out/r05_literal256_h256.log 2>&1
cp $(ls -t gpurun_out/r05_literal256_h256/*/*kernel_stats.csv | head -1) gpurun_out/r05_literal256_h256_kernel_stats.csv
tail -1 gpurun_out/r05_literal256_h256.log
python tools/time_literal256.py basic 8 1024 256 2>&1 | tail -1
tail -30 gpurun_out/r05f_prof_round.log
