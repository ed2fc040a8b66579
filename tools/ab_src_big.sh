# Same-box A/B of the working tree against saved copies of some sources, on the 1M-triangle scene:
#   (before the gpurun call)  mkdir -p tools/_old && git show HEAD:zdr_amd/csrc/X > tools/_old/X   for each file X
#   gpurun -- 'bash tools/ab_src_big.sh'     -> builds and times old, new, old, new (tools/run_big.py, 1024^2 spp $SPP)
cd $GRAFT_REPO_ROOT
SPP=${SPP:-32}
mkdir -p /tmp/new
for f in tools/_old/*; do cp zdr_amd/csrc/$(basename $f) /tmp/new/; done
for round in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp tools/_old/* zdr_amd/csrc/; else cp /tmp/new/* zdr_amd/csrc/; fi
    python -m zdr_amd.build --force > /dev/null 2>&1
    echo "== $v"; timeout -k 10 200 python tools/run_big.py --spp $SPP --iters 3 2>&1 | grep -E "^scene|^fwd|^bwd|image mean"
  done
done
cp /tmp/new/* zdr_amd/csrc/
python -m zdr_amd.build --force > /dev/null 2>&1
