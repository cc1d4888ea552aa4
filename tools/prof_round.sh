#!/bin/bash
# Round profile: (1) rocprofv3 kernel stats of the exact bench command, (2) PMC HBM traffic of the step kernel and of the
# PPO gradient kernel (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md §HBM).  Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r01}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench -- python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/${R}_bench.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench.log > gpurun_out/${R}_bench_n1.json
cp $(ls -t gpurun_out/${R}_bench/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_n1_kernel_stats.csv
# BASELINE configs[2]: Ball3D, 4096 envs, MLP 256x256 with bf16 MFMA operands -- kernel stats of the exact bench command
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_bench_ball3d_bf16 -- python bench.py --gpus 1 --steps 3 --warmup 1 --task ball3d --hidden 256 --mfma-dtype bf16 --no-cpu-baseline > gpurun_out/${R}_bench_ball3d_bf16.log 2>&1
grep -E "^\{" gpurun_out/${R}_bench_ball3d_bf16.log > gpurun_out/${R}_bench_ball3d_bf16_n1.json
cp $(ls -t gpurun_out/${R}_bench_ball3d_bf16/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_ball3d_bf16_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_step_$c -- python tools/env_sweep.py --tasks gridworld --sizes 4194304 --per-launch 1 --iters 2 > gpurun_out/${R}_pmc_step_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_grad_$c -- python tools/prof_grad.py > gpurun_out/${R}_pmc_grad_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_gradbf_$c -- python tools/prof_grad_bf16.py ball3d 256 bf16 > gpurun_out/${R}_pmc_gradbf_$c.log 2>&1
done
python - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def mean_counter(tag, counter, kernel_substr):
    f = glob.glob(f"gpurun_out/{R}_pmc_{tag}_{counter}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)
out = {}
for tag, sub, alg in (("step", "step_kernel<tma::GridTask, 3>", 54 * 4194304), ("grad", "ppo_grad_h64_kernel", None), ("gradbf", "ppo_grad_wide_bf_kernel", None)):
    fs, n1 = mean_counter(tag, "FETCH_SIZE", sub)
    ws, n2 = mean_counter(tag, "WRITE_SIZE", sub)
    d = {"kernel_match": sub, "dispatches": n1, "FETCH_SIZE_KB_mean": fs, "WRITE_SIZE_KB_mean": ws,
         "traffic_bytes_per_launch": (2 * fs + ws) * 1024,
         "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM; verified on refill_count_kernel: 8 B/env of dword loads read as exactly 1/2), WRITE_SIZE as is",
         "command": ("python tools/env_sweep.py --tasks gridworld --sizes 4194304 --per-launch 1 --iters 2" if tag == "step" else
                     "python tools/prof_grad.py" if tag == "grad" else "python tools/prof_grad_bf16.py ball3d 256 bf16")}
    if alg:
        d["algorithmic_bytes_per_launch"] = alg
        d["traffic_over_algorithmic"] = d["traffic_bytes_per_launch"] / alg
    out[tag] = d
    json.dump(d, open(f"gpurun_out/{R}_{tag}_kernel_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
python - "$R" <<'PY'
import csv, sys
R = sys.argv[1]
for r in list(csv.DictReader(open(f"gpurun_out/{R}_bench_n1_kernel_stats.csv")))[:12]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
cut -c1-400 gpurun_out/${R}_bench_n1.json
