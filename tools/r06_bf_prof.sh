cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for task in ${TASKS:-ball3d}; do
for n in ${NPIS:-128 132 136 140 144}; do
TMA_BF_NPI=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bfp_${task}_$n -- python tools/train_digest.py $task 256 bf16 131072 > /dev/null 2>&1
f=$(find gpurun_out/bfp_${task}_$n -name "*kernel_stats.csv" | head -1)
echo "$task n_pi $n: $(grep ppo_grad_wide_bf $f | awk -F'",' '{print $2}' | cut -d, -f1-3)"
done
done
