# usage: bash tools/ab_occupancy.sh "<flags variant 1>" ...   — cbox path / path + environment / direct passes per build variant
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== flags: [$v]"
  ZDR_KERNEL_FLAGS="$v" python -m zdr_amd.build --force > /dev/null 2>gpurun_out/ab_build.log || { tail -5 gpurun_out/ab_build.log; continue; }
  echo "path:";   timeout -k 10 120 python tools/run_pass.py --which both --iters 4 2>&1 | grep -E "fwd|bwd"
  echo "path + environment map:"; timeout -k 10 120 python tools/run_pass.py --which both --iters 4 --env 2>&1 | grep -E "fwd|bwd"
  echo "direct spp 64:"; timeout -k 10 120 python tools/run_pass.py --which both --iters 6 --integrator direct --spp 64 2>&1 | grep -E "fwd|bwd"
done
