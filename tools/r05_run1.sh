#!/bin/bash
# round 5 GPU pass: tests, step-kernel A/B (round-3 library vs this build on ONE box), data-parallel chain A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05a_pytest.log
python tools/step_ab.py --lib tools/bin/libtma_hip_r03.so --lib three-mlagents_amd/csrc/libtma_hip.so > gpurun_out/r05a_step_ab.jsonl 2>&1
TMA_STEP_THREADS=64 python tools/step_ab.py --lib three-mlagents_amd/csrc/libtma_hip.so --rounds 2 > gpurun_out/r05a_step_t64.jsonl 2>&1
TMA_STEP_THREADS=128 python tools/step_ab.py --lib three-mlagents_amd/csrc/libtma_hip.so --rounds 2 > gpurun_out/r05a_step_t128.jsonl 2>&1
python tools/step_ab.py --lib tools/bin/libtma_hip_r03.so --lib three-mlagents_amd/csrc/libtma_hip.so --log-cap 1048576 --rounds 2 > gpurun_out/r05a_step_ab_log.jsonl 2>&1
for i in 1 2; do
TMA_DP_PATH=1 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05a_dp_fold_$i.json
TMA_DP_PATH=1 TMA_DP_NO_FOLD=1 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05a_dp_old_$i.json
done
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05a_local.json
grep -h median gpurun_out/r05a_step*.jsonl
python - <<'PY'
import json
for f in ("dp_fold_1","dp_old_1","dp_fold_2","dp_old_2","local"):
    try:
        d=json.loads(open(f"gpurun_out/r05a_{f}.json").read()); print(f, d["value"], d["update_ms"], d["update_ms"]/320*1e3, "us per minibatch")
    except Exception as e: print(f, "ERR", e)
PY
