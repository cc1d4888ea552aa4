mkdir -p gpurun_out
( timeout 500 python -m pytest "tests/test_dist_gpu.py::test_peer_exchange_protocol_at_node_world_sizes_in_one_process" "tests/test_dist_gpu.py::test_bench_two_ranks_end_to_end_on_one_gpu" -q --durations=8 2>&1 | grep -v 'amdgpu.ids\|socket.cpp\|Gloo' | tail -30 ) > gpurun_out/r06_dist3.log
cat gpurun_out/r06_dist3.log
