cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
tail -2 gpurun_out/prof_bench.log | cut -c1-600
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>7s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
