cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_bf_$tag -- python tools/prof_grad_bf16.py "$@" > gpurun_out/pmc_bf_$tag.log 2>&1
  python - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_bf_{tag}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counters for", tag, open(f"gpurun_out/pmc_bf_{tag}.log").read()[-600:]); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "ppo_grad_wide" in r["Kernel_Name"] or "ppo_grad_split3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"  {k:32s} mean={sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
