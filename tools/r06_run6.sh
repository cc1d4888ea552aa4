mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'amdgpu.ids\|socket.cpp\|Gloo' | tail -25 ) > gpurun_out/r06_gputest.log
( TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_ticks.so timeout 300 python tools/h64_ticks.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_h64_ticks.txt
cat gpurun_out/r06_gputest.log gpurun_out/r06_h64_ticks.txt
