# usage: bash tools/ab_big.sh "<flags variant 1>" "<flags variant 2>" ...  — each variant is rebuilt on the box, then the
# 1M-triangle scene (BASELINE configs[4]) is timed: path forward + backward at 1024^2 spp $SPP (default 32) and, with
# TRACE=1, the traversal-only benchmark.  Variants run twice, interleaved, so drift of the box shows.  A variant that
# does not finish within 60 s ends the script (never keep launching on a GPU that has just hung a kernel).
cd $GRAFT_REPO_ROOT
SPP=${SPP:-32}
for round in 1 2; do
  for v in "$@"; do
    echo "== round $round flags: [$v]"
    ZDR_KERNEL_FLAGS="$v" python -m zdr_amd.build --force > /dev/null 2>gpurun_out/ab_build.log || { tail -5 gpurun_out/ab_build.log; continue; }
    timeout -k 5 60 python tools/run_big.py --spp $SPP --iters 3 > gpurun_out/ab_run.log 2>&1 || { echo "run failed or timed out: stopping"; tail -3 gpurun_out/ab_run.log; python -m zdr_amd.build --force > /dev/null 2>&1; exit 1; }
    grep -E "^fwd|^bwd|image mean" gpurun_out/ab_run.log
    if [ "$TRACE" = "1" ] && [ $round = 1 ]; then timeout -k 5 100 python tools/trace_bench.py 2>&1 | grep -E "Mrays"; fi
  done
done
python -m zdr_amd.build --force > /dev/null 2>&1
