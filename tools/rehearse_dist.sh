# 2 ranks on the one GPU of this box over gloo: exercises bench.py's multi-rank path end to end
cd $GRAFT_REPO_ROOT
for shard in seeds rows samples; do
  echo "== shard=$shard"
  ZDR_DIST_BACKEND=gloo ZDR_SHARE_DEVICE=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --shard $shard 2>&1 | grep -E '^\{|Error|error' | cut -c1-700
done
