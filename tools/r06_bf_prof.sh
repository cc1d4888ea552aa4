cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for task in ball3d push; do
for tag in prev new; do
lib=three-mlagents_amd/csrc/libtma_hip.so; [ $tag = prev ] && lib=tools/bin/libtma_hip_prev.so
TMA_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bfp_${task}_$tag -- python tools/train_digest.py $task 256 bf16 131072 > /dev/null 2>&1
f=$(find gpurun_out/bfp_${task}_$tag -name "*kernel_stats.csv" | head -1)
echo "$task $tag: $(grep ppo_grad_wide_bf $f | awk -F'",' '{print $2}' | cut -d, -f1-3)"
done
done
