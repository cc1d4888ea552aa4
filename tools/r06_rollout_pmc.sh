#!/bin/bash
# round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the headline rollout kernel (rollout_chunk4_h64_kernel<GridTask>,
# 4096 envs, 512 vector steps per launch) -> gpurun_out/r06_rollout_kernel_pmc.json.   usage (GPU box): bash tools/r06_rollout_pmc.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r06_pmc_rollout_$i -- python tools/time_rollout.py gridworld 4096 1024 64 f32 > gpurun_out/r06_pmc_rollout_$i.log 2>&1
done
python - <<'PY'
import collections, csv, glob, json
agg, steps = collections.defaultdict(list), None
for i in (1, 2, 3, 4):
    for f in glob.glob(f"gpurun_out/r06_pmc_rollout_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_chunk4_h64_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
# full-window launches only (the ring window is 512 vector steps; a rollout of 1024 steps is two of them): the largest values of each counter
m = {}
for k, v in agg.items():
    top = sorted(v)[len(v) // 2:]
    m[k] = sum(top) / len(top)
out = {"kernel_match": "rollout_chunk4_h64_kernel<GridTask>", "command": "python tools/time_rollout.py gridworld 4096 1024 64 f32  (4096 envs; launches of 512 vector steps)",
       "counters_mean_per_launch": m, "vector_steps_per_launch": 512}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out["traffic_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    out["traffic_bytes_per_vector_step"] = out["traffic_bytes_per_launch"] / 512
    out["algorithmic_bytes_per_vector_step"] = 4096 * (4 * 4 + 4 + 16)  # (4 D + 4 A' + 16) per env-step, GridWorld D = 4
    out["correction"] = "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE as is; KB units"
if m.get("SQ_INSTS_MFMA") and m.get("SQ_INSTS_VALU"):
    out["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("SQ_BUSY_CYCLES"):
    out["kernel_cycles"] = m["SQ_BUSY_CYCLES"] / 32      # summed over the 32 shader engines (tools/prof_round.sh)
    out["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / out["kernel_cycles"]   # over the 1024 SIMDs
json.dump(out, open("gpurun_out/r06_rollout_kernel_pmc.json", "w"), indent=1)
print(json.dumps(out)[:900])
PY
