#!/bin/bash
# SQ and traffic counters of the 256-wide persistent epoch kernel (one launch = 4096 optimizer steps on GridWorld 4096 x 256, 256 x 256 f32)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
i=0
for set in "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r06_h256p_pmc$i -- python tools/time_literal256.py gridworld 4096 256 256 > gpurun_out/r06_h256p_pmc$i.log 2>&1
done
python - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(list)
for i in range(1, 6):
    for f in glob.glob(f"gpurun_out/r06_h256p_pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ppo_epoch_h256p_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
steps = 4096
out = {"kernel_match": "ppo_epoch_h256p_kernel", "command": "python tools/time_literal256.py gridworld 4096 256 256  (one launch = 4096 optimizer steps; 64 of the 256 CUs hold roles)",
       "counters_mean_per_launch": m, "optimizer_steps_per_launch": steps}
if m.get("SQ_BUSY_CYCLES"):
    out["kernel_cycles"] = m["SQ_BUSY_CYCLES"] / 32  # (summed over the 32 shader engines; only 8 of them -- two XCDs -- are busy for the whole launch: see note)
if m.get("SQ_INSTS_MFMA"):
    out["mfma_instructions_per_step"] = m["SQ_INSTS_MFMA"] / steps
    out["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_VALU") else None
if m.get("SQ_INSTS_VALU_MFMA_MOPS_F32"):
    out["issued_f32_mfma_flops_per_step"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / steps
if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
    out["mfma_busy_cycles_per_role_simd_per_step"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (64 * 4) / steps  # 64 role CUs x 4 SIMDs
if m.get("SQ_LDS_BANK_CONFLICT") and m.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
if m.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if m.get(k):
            out[k.lower() + "_frac_of_wave_cycles"] = m[k] / m["SQ_WAVE_CYCLES"]
if m.get("FETCH_SIZE") is not None and m.get("WRITE_SIZE") is not None:
    out["fabric_traffic_bytes_per_step"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / steps
    out["traffic_note"] = "FETCH_SIZE x 2 + WRITE_SIZE (MI355X_MICROARCH.md); the exchanges live in the two XCDs' L2s, what crosses the fabric is rollout rows, write-backs and the cross-XCD granules"
json.dump(out, open("gpurun_out/r06_h256p_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters_mean_per_launch"}, indent=0))
PY
