# usage: bash tools/ab.sh "<flags variant 1>" "<flags variant 2>" ...   (each rebuilt on the box, then fwd+bwd timed)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== flags: [$v]"
  ZDR_KERNEL_FLAGS="$v" python -m zdr_amd.build --force > /dev/null 2>gpurun_out/ab_build.log || { tail -5 gpurun_out/ab_build.log; continue; }
  timeout -k 10 120 python tools/run_pass.py --which both --iters 4 2>&1 | grep -E "fwd|bwd"
done
