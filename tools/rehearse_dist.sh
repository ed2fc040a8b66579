# 2 ranks on the one GPU of this box over gloo: exercises bench.py's own multi-rank launch (python bench.py --gpus 2 spawns
# torch.distributed.run as a child) and every shard mode end to end.  A functional rehearsal, not a scaling number.
cd $GRAFT_REPO_ROOT
for shard in tiles rows samples seeds; do
  echo "== python bench.py --gpus 2 --shard $shard   (ZDR_DIST_BACKEND=gloo ZDR_SHARE_DEVICE=1, c4 at spp 64)"
  ZDR_DIST_BACKEND=gloo ZDR_SHARE_DEVICE=1 timeout -k 10 200 python bench.py --gpus 2 --steps 3 --warmup 1 --spp 64 --shard $shard 2>&1 | grep -E '^\{|Error|error'
done
