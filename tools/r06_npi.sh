for n in 120 124 128 132 136 140; do echo "n_pi $n"; TMA_BF_NPI=$n python tools/time_grad.py ball3d 256 bf16 131072 2>&1 | grep "grad call"; done
echo prev; TMA_LIB_PATH=tools/bin/libtma_hip_prev.so python tools/time_grad.py ball3d 256 bf16 131072 push 256 bf16 131072 2>&1 | grep "grad call"
