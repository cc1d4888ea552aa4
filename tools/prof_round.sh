#!/bin/bash
# Round profile (run on the GPU box through gpurun): outputs under gpurun_out/, the summaries to commit are copied to profiles/ by hand.
#  (1) timeout 300 rocprofv3 --kernel-trace --stats of the headline bench command (the timed region only: --no-extras --no-cpu-baseline, so that
#      the per-kernel averages are those of the headline workload and not mixed with the batch-256 / other-config legs, which launch the same kernels
#      at other sizes) and of the Ball3D 256x256 bf16 command;
#  (2) HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes; MI355X_MICROARCH.md §HBM) of the step kernel and the gradient kernels;
#  (3) SQ counters (MFMA busy cycles / instruction counts / wait cycles / LDS conflicts) of the three gradient kernels.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench.log > gpurun_out/${R}_bench_n1.json
cp $(ls -t gpurun_out/${R}_bench/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_n1_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench_ball3d_bf16 -- python bench.py --gpus 1 --steps 3 --warmup 1 --task ball3d --hidden 256 --mfma-dtype bf16 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_ball3d_bf16.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench_ball3d_bf16.log > gpurun_out/${R}_bench_ball3d_bf16_n1.json
cp $(ls -t gpurun_out/${R}_bench_ball3d_bf16/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_ball3d_bf16_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench_crawler_bf16 -- python bench.py --gpus 1 --steps 2 --warmup 1 --task crawler --n-envs 2048 --n-steps 2048 --hidden 256 --mfma-dtype bf16 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_crawler_bf16.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench_crawler_bf16.log > gpurun_out/${R}_bench_crawler_bf16_n1.json
cp $(ls -t gpurun_out/${R}_bench_crawler_bf16/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_crawler_bf16_kernel_stats.csv
# GridWorld 4096 envs with the reference's default net, MLP(256, 256) f32 (second half)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench_gridworld_f32w -- python bench.py --gpus 1 --steps 2 --warmup 1 --hidden 256 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_gridworld_f32w.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench_gridworld_f32w.log > gpurun_out/${R}_bench_gridworld_f32w_n1.json
cp $(ls -t gpurun_out/${R}_bench_gridworld_f32w/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_gridworld_f32w_kernel_stats.csv
# the reference's literal batch_size = 256: one persistent launch per epoch (ppo_epoch_h64p_kernel), 4096 envs x 256 steps = 4096 optimizer steps per launch
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_literal256 -- python tools/time_epoch256.py 4096 256 > gpurun_out/${R}_literal256.log 2>&1
cp $(ls -t gpurun_out/${R}_literal256/*/*kernel_stats.csv | head -1) gpurun_out/${R}_literal256_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_step_$c -- python tools/env_sweep.py --tasks gridworld --sizes 4194304 --per-launch 1 --iters 2 > gpurun_out/${R}_pmc_step_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_grad_$c -- python tools/prof_grad.py > gpurun_out/${R}_pmc_grad_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradbf_$c -- python tools/prof_grad_bf16.py ball3d 256 bf16 > gpurun_out/${R}_pmc_gradbf_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradbf_push_$c -- python tools/prof_grad_bf16.py push 256 bf16 > gpurun_out/${R}_pmc_gradbf_push_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradbf_crawler_$c -- python tools/prof_grad_bf16.py crawler 256 bf16 > gpurun_out/${R}_pmc_gradbf_crawler_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradwide_basic_$c -- python tools/prof_grad_bf16.py basic 256 f32 8 256 > gpurun_out/${R}_pmc_gradwide_basic_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradwide_gridworld_$c -- python tools/prof_grad_bf16.py gridworld 256 f32 > gpurun_out/${R}_pmc_gradwide_gridworld_$c.log 2>&1
done
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
i=0
for set in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${R}_sq${i}_h64 -- python tools/prof_grad.py > gpurun_out/${R}_sq${i}_h64.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${R}_sq${i}_bf16 -- python tools/prof_grad_bf16.py ball3d 256 bf16 > gpurun_out/${R}_sq${i}_bf16.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${R}_sq${i}_f32w -- python tools/prof_grad_bf16.py ball3d 256 f32 > gpurun_out/${R}_sq${i}_f32w.log 2>&1
done
python - "$R" <<'PY'
import collections, csv, glob, json, sys
R = sys.argv[1]
def counters(dirname, kernel_substr):
    f = glob.glob(f"gpurun_out/{dirname}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, (len(next(iter(agg.values()))) if agg else 0)
out = {}
for tag, sub, alg, cmd in (("step", "step_kernel<tma::GridTask, 3>", 54 * 4194304, "python tools/env_sweep.py --tasks gridworld --sizes 4194304 --per-launch 1 --iters 2"),
                           ("grad", "ppo_grad_h64_kernel", None, "python tools/prof_grad.py"),
                           ("gradbf", "ppo_grad_wide_bf_kernel", None, "python tools/prof_grad_bf16.py ball3d 256 bf16"),
                           ("gradbf_push", "ppo_grad_wide_bf_kernel", None, "python tools/prof_grad_bf16.py push 256 bf16"),
                           ("gradbf_crawler", "ppo_grad_wide_bf_kernel", None, "python tools/prof_grad_bf16.py crawler 256 bf16  (both launches of the two-pass layout: per-launch mean)"),
                           ("gradwide_basic", "ppo_grad_wide_kernel", None, "python tools/prof_grad_bf16.py basic 256 f32 8 256  (256 samples per launch)"),
                           ("gradwide_gridworld", "ppo_grad_wide_kernel", None, "python tools/prof_grad_bf16.py gridworld 256 f32  (the reference's default net on the headline env, 131072 samples per launch)")):
    fs, n1 = counters(f"{R}_pmc_{tag}_FETCH_SIZE", sub)
    ws, n2 = counters(f"{R}_pmc_{tag}_WRITE_SIZE", sub)
    if not fs or not ws:
        continue
    d = {"kernel_match": sub, "dispatches": n1, "FETCH_SIZE_KB_mean": fs["FETCH_SIZE"], "WRITE_SIZE_KB_mean": ws["WRITE_SIZE"],
         "traffic_bytes_per_launch": (2 * fs["FETCH_SIZE"] + ws["WRITE_SIZE"]) * 1024,
         "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM), WRITE_SIZE as is; the counters are L2 fabric requests: Infinity-Cache hits are included",
         "command": cmd}
    if alg:
        d["algorithmic_bytes_per_launch"] = alg
        d["traffic_over_algorithmic"] = d["traffic_bytes_per_launch"] / alg
    out[tag] = d
    json.dump(d, open(f"gpurun_out/{R}_{tag}_kernel_pmc.json", "w"), indent=1)
sq = {}
for tag, sub, cmd in (("h64", "ppo_grad_h64_kernel", "python tools/prof_grad.py  (GridWorld 64x64 f32, 131072 samples per launch)"),
                      ("bf16", "ppo_grad_wide_bf_kernel", "python tools/prof_grad_bf16.py ball3d 256 bf16  (131072 samples per launch)"),
                      ("f32w", "ppo_grad_wide_kernel", "python tools/prof_grad_bf16.py ball3d 256 f32  (131072 samples per launch)")):
    c = {}
    for i in (1, 2, 3):
        cc, n = counters(f"{R}_sq{i}_{tag}", sub)
        c.update(cc)
    if not c:
        continue
    simds = 256 * 4
    d = {"kernel_match": sub, "command": cmd, "counters_mean_per_launch": c}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles)
        kernel_cycles = c["SQ_BUSY_CYCLES"] / 32
        d["kernel_cycles"] = kernel_cycles
        d["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / kernel_cycles
        d["mfma_busy_cycles_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simds
    if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
        d["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in c:
                d[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    sq[tag] = d
json.dump(sq, open(f"gpurun_out/{R}_grad_kernels_sq_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
print(json.dumps({k: {x: y for x, y in v.items() if x != "counters_mean_per_launch"} for k, v in sq.items()}, indent=1))
PY
python - "$R" <<'PY'
import csv, sys
R = sys.argv[1]
for name in (f"gpurun_out/{R}_bench_n1_kernel_stats.csv", f"gpurun_out/{R}_bench_ball3d_bf16_kernel_stats.csv", f"gpurun_out/{R}_bench_crawler_bf16_kernel_stats.csv", f"gpurun_out/{R}_bench_gridworld_f32w_kernel_stats.csv",
             f"gpurun_out/{R}_literal256_kernel_stats.csv"):
    print(name)
    for r in list(csv.DictReader(open(name)))[:8]:
        print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
cut -c1-300 gpurun_out/${R}_bench_n1.json
