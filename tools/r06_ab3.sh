# A/B of two libraries on ONE box, h256p only: timing + parameter digest (a numerics-neutral change must leave the digest alone) + phase ticks
mkdir -p gpurun_out
OLD=${1:-tools/bin/libtma_hip_prev.so}
NEW=three-mlagents_amd/csrc/libtma_hip.so
: > gpurun_out/ab3.log
for i in 1 2; do
for lib in $OLD $NEW; do
echo "== $lib" >> gpurun_out/ab3.log
TMA_LIB_PATH=$lib timeout 300 python tools/time_literal256.py gridworld 4096 256 256 2>&1 | grep "optimizer steps" >> gpurun_out/ab3.log
TMA_LIB_PATH=$lib timeout 300 python tools/time_literal256.py basic 4096 256 256 2>&1 | grep "optimizer steps" >> gpurun_out/ab3.log
done
done
TMA_LIB_PATH=$NEW timeout 300 python tools/h256p_ticks.py >> gpurun_out/ab3.log 2>&1
timeout 900 python -m pytest tests/test_h256p_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/ab3.log
cat gpurun_out/ab3.log
