#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of ppo_grad_h64_kernel in the headline bench, with the sample records (default) and gathering from the planes (TMA_NO_PACKED=1).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in planes packed; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $mode = planes ]; then export TMA_NO_PACKED=1; else unset TMA_NO_PACKED; fi
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${mode}_$c -- python bench.py --gpus 1 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/pmc_${mode}_$c.log 2>&1
  done
done
python - <<'PY'
import collections, csv, glob, json
out = {}
for mode in ("planes", "packed"):
    d = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc_{mode}_{c}/**/*counter_collection.csv", recursive=True)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            for key in ("ppo_grad_h64_kernel", "adv_partial_kernel", "pack_samples_kernel"):
                if key in r["Kernel_Name"] and r["Counter_Name"] == c:
                    agg[key].append(float(r["Counter_Value"]))
        for key, v in agg.items():
            d.setdefault(key, {})[c + "_KB_mean"] = sum(v) / len(v)
            d[key]["dispatches"] = len(v)
    for key, v in d.items():
        if "FETCH_SIZE_KB_mean" in v and "WRITE_SIZE_KB_mean" in v:
            v["traffic_MB_per_launch"] = (2 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]) * 1024 / 1e6  # FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md)
    out[mode] = d
json.dump(out, open("gpurun_out/r02_packed_records_pmc.json", "w"), indent=1)
g = out["packed"]["ppo_grad_h64_kernel"]
json.dump({"kernel_match": "ppo_grad_h64_kernel", "dispatches": g["dispatches"], "FETCH_SIZE_KB_mean": g["FETCH_SIZE_KB_mean"], "WRITE_SIZE_KB_mean": g["WRITE_SIZE_KB_mean"],
           "traffic_bytes_per_launch": g["traffic_MB_per_launch"] * 1e6,
           "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section), WRITE_SIZE as is; the counters are L2 fabric requests: Infinity-Cache hits are included",
           "command": "python bench.py --gpus 1 --steps 1 --warmup 1 --no-extras --no-cpu-baseline  (tools/pmc_packed.sh; sample records on, the default)"},
          open("gpurun_out/r02_grad_kernel_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
