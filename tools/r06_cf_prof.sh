cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in crawler ant; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cf_$t -- python tools/train_digest.py $t 256 f32 131072 > /dev/null 2>&1
f=$(find gpurun_out/cf_$t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage'])>0.3: print(r['Name'][:75], r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
PY
done
