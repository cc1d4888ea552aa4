#!/bin/bash
# round 5: the data-parallel minibatch chain at world size 1 on ONE box -- no collective (callback no-op), RCCL's launch in the chain,
# the peer exchange fused into slab_reduce / sum-of-squares, and the exchange as launches of its own
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r05d_$name.err | tail -1 > gpurun_out/r05d_$name.json; }
for i in 1 2; do
run none_$i TMA_DP_PATH=1
run rccl_$i TMA_DP_PATH=1 TMA_NATIVE_RCCL=1 TMA_P2P=0
run p2p_$i TMA_DP_PATH=1 TMA_NATIVE_RCCL=1 TMA_P2P=1
run p2pnf_$i TMA_DP_PATH=1 TMA_NATIVE_RCCL=1 TMA_P2P=1 TMA_P2P_NO_FUSE=1
done
run local TMA_NONE=1
python - <<'PY'
import json
for f in ("none_1","rccl_1","p2p_1","p2pnf_1","none_2","rccl_2","p2p_2","p2pnf_2","local"):
    try:
        d=json.loads(open(f"gpurun_out/r05d_{f}.json").read()); print(f, round(d["value"]/1e6,2), "M", d["update_ms"], round(d["update_ms"]/320*1e3,2), "us per minibatch")
    except Exception as e: print(f, "ERR", e); print(open(f"gpurun_out/r05d_{f}.err").read()[-800:])
PY
grep -h "peer exchange" gpurun_out/r05d_p2p_1.err | head -3
