# usage: bash tools/ab_pmc.sh "<flags 1>" "<flags 2>" ...   — per variant (rebuilt on the box): wall time of the backward pass, then one
# counter pass (cycles, waits, clock) of the same pass.  Variants run twice, interleaved.
cd $GRAFT_REPO_ROOT
n=0
for round in 1 2; do
for v in "$@"; do
  n=$((n+1))
  echo "== flags: [$v]"
  ZDR_KERNEL_FLAGS="$v" python -m zdr_amd.build --force > /dev/null 2>gpurun_out/ab_build.log || { tail -5 gpurun_out/ab_build.log; continue; }
  timeout -k 10 120 python tools/run_pass.py --which bwd --iters 4 2>&1 | grep -E "bwd"
  rm -rf gpurun_out/pmcc_abp${n}_*
  bash tools/pmc_custom.sh abp$n "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY" --which bwd 2>&1 | tail -1 | cut -c60-400
done
done
python -m zdr_amd.build --force > /dev/null 2>&1
