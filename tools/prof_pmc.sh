# HBM traffic of the step kernel from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python tools/env_sweep.py --tasks gridworld --sizes 4194304 --per-launch 1 --iters 2 > gpurun_out/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(c, "no counter file", glob.glob(f"gpurun_out/pmc_{c}/**/*", recursive=True)[:10]); continue
    rows = list(csv.DictReader(open(f[0])))
    print(c, "columns:", list(rows[0].keys()))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"  {k:60s} n={len(v):4d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
PY
