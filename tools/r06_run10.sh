mkdir -p gpurun_out
( timeout 900 python tools/threshold_runs.py --tasks gridworld --schedules literal --seeds 1,2,3,4,5,6,7 --out gpurun_out/r06_thr_gridworld_literal_b.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for ln in sys.stdin:
    try:
        d = json.loads(ln); print(d['seed'], d['final_eval_mean'], d['reached'], d['first_eval_at_threshold'] is not None)
    except Exception: pass
" ) > gpurun_out/r06_thr_b.log
cat gpurun_out/r06_thr_b.log
