# usage: bash tools/sweep_big.sh "VAR=val ..." ...   (1M-triangle path workload, no rebuild)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== env: [$v]"
  env $v timeout -k 10 200 python tools/run_big.py --res 1024 --spp 32 --iters 2 2>&1 | grep -E "^fwd|^bwd"
done
