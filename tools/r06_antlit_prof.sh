cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/antlit -- python bench.py --task ant --n-envs 8 --n-steps 1024 --hidden 256 --batch-size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find gpurun_out/antlit -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage'])>0.5: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
PY
