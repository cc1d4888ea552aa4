#!/bin/bash
# kernel statistics of the fused rollout alone (tools/time_rollout.py): which launches share the rollout stream with the chunk kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_rollout -- python tools/time_rollout.py ${@:-gridworld 4096 1024 64 f32} > gpurun_out/r03_rollout.log 2>&1
tail -3 gpurun_out/r03_rollout.log
cp $(ls -t gpurun_out/r03_rollout/*/*kernel_stats.csv | head -1) gpurun_out/r03_rollout_kernel_stats.csv
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r03_rollout_kernel_stats.csv")))[:14]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
