#!/bin/bash
# round 6, final measurement pass on ONE box: the driver's default bench command (line + extras), the round's rocprof / PMC summaries
# (tools/prof_round.sh), the 256-wide literal-batch epoch (persistent kernel and launch path), the three-term split kernel's SQ counters,
# the phase ticks of the two persistent kernels' neighbours, the reward-threshold runs.   usage: r06_final.sh [quick]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
MODE=${1:-full}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (a) the reference's literal batch_size = 256 on its 256 x 256 net: ONE persistent launch per epoch (ppo_epoch_h256p_kernel) ...
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_literal256_h256 -- python tools/time_literal256.py gridworld 4096 256 256 > gpurun_out/r06_literal256_h256.log 2>&1
cp $(ls -t gpurun_out/r06_literal256_h256/*/*kernel_stats.csv | head -1) gpurun_out/r06_literal256_h256_kernel_stats.csv
# ... and the same epoch as per-minibatch launches (TMA_NO_PERSIST=1: round 5's path, the fallback)
TMA_NO_PERSIST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_literal256_h256_launches -- python tools/time_literal256.py gridworld 4096 256 256 > gpurun_out/r06_literal256_h256_launches.log 2>&1
cp $(ls -t gpurun_out/r06_literal256_h256_launches/*/*kernel_stats.csv | head -1) gpurun_out/r06_literal256_h256_launches_kernel_stats.csv
( grep "optimizer steps" gpurun_out/r06_literal256_h256.log; grep "optimizer steps" gpurun_out/r06_literal256_h256_launches.log; python tools/time_literal256.py basic 8 1024 256 2>&1 | tail -1; TMA_NO_PERSIST=1 python tools/time_literal256.py basic 8 1024 256 2>&1 | tail -1 ) > gpurun_out/r06_literal256_time.txt
python tools/h256p_ticks.py gridworld 1024 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_h256p_ticks.txt
[ -f three-mlagents_amd/csrc/libtma_hip_ticks.so ] && TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_ticks.so python tools/h64_ticks.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_h64_ticks.txt
cat gpurun_out/r06_literal256_time.txt gpurun_out/r06_h256p_ticks.txt
[ "$MODE" = "quick" ] && exit 0
# (b) reward thresholds: every case once, the literal GridWorld schedule on five seeds
python tools/threshold_runs.py --out gpurun_out/r06_thresholds_all.json > gpurun_out/r06_thresholds.log 2>&1
python tools/threshold_runs.py --tasks gridworld --schedules literal --seeds 1,2,3,4,5 --out gpurun_out/r06_thresholds_gridworld_literal_seeds.json >> gpurun_out/r06_thresholds.log 2>&1
# (c) the driver's default command
t0=$(date +%s)
python bench.py > gpurun_out/r06_bench_line_n1.json 2> gpurun_out/r06_bench.err
echo "default bench: $(( $(date +%s) - t0 )) s, line $(wc -c < gpurun_out/r06_bench_line_n1.json) bytes"
cp bench_extras.json gpurun_out/r06_bench_extras_n1.json 2>/dev/null
# (d) kernel statistics, HBM traffic and SQ counters of the gradient / step kernels
bash tools/prof_round.sh r06 > gpurun_out/r06_prof_round.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_gridworld_bf16x3 -- python bench.py --gpus 1 --steps 2 --warmup 1 --hidden 256 --mfma-dtype bf16x3 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_gridworld_bf16x3.log 2>&1
cp $(ls -t gpurun_out/r06_bench_gridworld_bf16x3/*/*kernel_stats.csv | head -1) gpurun_out/r06_bench_gridworld_bf16x3_kernel_stats.csv
# (e) the three-term split kernel's SQ counters (what it ISSUES on the bf16 pipe: its roofline against the bf16 peak)
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=0
for set in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r06_sq${i}_split3 -- python tools/prof_grad_bf16.py gridworld 256 bf16x3 > gpurun_out/r06_sq${i}_split3.log 2>&1
done
python - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(list)
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/r06_sq{i}_split3/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ppo_grad_split3_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"kernel_match": "ppo_grad_split3_kernel", "command": "python tools/prof_grad_bf16.py gridworld 256 bf16x3  (131072 samples per launch)", "counters_mean_per_launch": m}
if m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
    # MOPS counters are in units of 512 operations (MI355X_MICROARCH.md); issued bf16 flops per launch = MOPS x 512
    out["issued_bf16_flops_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    out["issued_over_f32_equivalent"] = out["issued_bf16_flops_per_launch"] / (131072 * 807936)
if m.get("SQ_INSTS_MFMA") and m.get("SQ_INSTS_VALU"):
    out["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("SQ_BUSY_CYCLES"):
    # SQ_BUSY_CYCLES is summed over the 32 shader engines; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (tools/prof_round.sh)
    out["kernel_cycles"] = m["SQ_BUSY_CYCLES"] / 32
    out["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / out["kernel_cycles"]
if m.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if m.get(k):
            out[k.lower() + "_frac_of_wave_cycles"] = m[k] / m["SQ_WAVE_CYCLES"]
if m.get("SQ_LDS_BANK_CONFLICT") and m.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open("gpurun_out/r06_gradsplit3_sq_pmc.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
tail -5 gpurun_out/r06_prof_round.log
