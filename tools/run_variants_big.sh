# Times prebuilt kernel variants (tools/build_variants.sh) on the GPU box, twice, interleaved: the 1M-triangle scene, path 1024^2 spp $SPP.
#   gpurun -- 'SPP=32 bash tools/run_variants_big.sh base w5 w7 ...'      (TRACE=1: also the traversal-only ray rates, first round)
# RUN="python tools/run_pass.py --which both --iters 6" times another workload instead (the cbox bench pass).
cd $GRAFT_REPO_ROOT
SPP=${SPP:-32}
cp zdr_amd/csrc/libzdr_hip.so /tmp/libzdr_hip_tree.so
for round in 1 2; do
  for v in "$@"; do
    cp tools/_variants/libzdr_hip_$v.so zdr_amd/csrc/libzdr_hip.so || { echo "no such variant $v"; continue; }
    echo "== round $round variant $v"
    timeout -k 5 90 ${RUN:-python tools/run_big.py --spp $SPP --iters 3} > gpurun_out/ab_run.log 2>&1 || { echo "run failed or timed out: stopping"; tail -3 gpurun_out/ab_run.log; cp /tmp/libzdr_hip_tree.so zdr_amd/csrc/libzdr_hip.so; exit 1; }
    grep -E "^fwd|^bwd|image mean" gpurun_out/ab_run.log
    if [ "$TRACE" = "1" ] && [ $round = 1 ]; then timeout -k 5 100 python tools/trace_bench.py 2>&1 | grep -E "Mrays"; fi
  done
done
cp /tmp/libzdr_hip_tree.so zdr_amd/csrc/libzdr_hip.so
