mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'amdgpu.ids\|socket.cpp' | tail -25 ) > gpurun_out/r06_gputest.log
( timeout 1500 python bench.py 2> gpurun_out/r06_bench_stderr.log | tail -1 ) > gpurun_out/r06_bench_line_n1.json
cp bench_extras.json gpurun_out/r06_bench_extras_n1.json 2>/dev/null
cat gpurun_out/r06_gputest.log; tail -40 gpurun_out/r06_bench_stderr.log; cat gpurun_out/r06_bench_line_n1.json
