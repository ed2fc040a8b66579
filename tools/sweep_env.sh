# usage: bash tools/sweep_env.sh "VAR=val VAR2=val" "VAR=val" ...   (no rebuild; each variant timed fwd+bwd)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== env: [$v]"
  env $v timeout -k 10 120 python tools/run_pass.py --which both --iters 4 2>&1 | grep -E "fwd|bwd"
done
