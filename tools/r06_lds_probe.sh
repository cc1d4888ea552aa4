#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of LDS read streams that are conflict-free by construction (tools/lds_counter_probe.hip)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/lds_probe -- tools/bin/lds_counter_probe > gpurun_out/lds_probe.log 2>&1
python - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/lds_probe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for k, m in agg.items():
    if m.get("SQ_LDS_IDX_ACTIVE"):
        out[k] = {"conflict_cycles_over_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], "lds_active_cycles_per_instruction": m["SQ_LDS_IDX_ACTIVE"] / m["SQ_INSTS_LDS"], **m}
json.dump(out, open("gpurun_out/r06_lds_counter_probe.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k:16s} conflict / active = {v['conflict_cycles_over_lds_active']:.3f}   active cycles per LDS instruction = {v['lds_active_cycles_per_instruction']:.2f}")
PY
