set -e
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','fwd_ms','bwd_ms','fwd_msamples_s','bwd_msamples_s')})"; }
run ZDR_X=1
run ZDR_DEBUG_NO_SCATTER=1
run ZDR_TARGET_WAVES=4096
run ZDR_TARGET_WAVES=8192
run ZDR_TARGET_WAVES=32768
run ZDR_TARGET_WAVES=65536
