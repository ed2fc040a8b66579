# Builds kernel VARIANTS here (the CPU container cross-compiles gfx950) so that a gpurun call only has to time them — builds on the GPU box
# cost GPU-minutes.  usage: bash tools/build_variants.sh name1="<flags or @patchfile>" name2=...
#   flags      ZDR_KERNEL_FLAGS of the variant, e.g.  w5="-DZDR_MIN_WAVES_BVH=5"
#   @file      a patch (patch -p1) applied to a copy of the tree, e.g.  anyhit=@profiles/r4_exp_anyhit_unordered.patch
# -> tools/_variants/libzdr_hip_<name>.so (git-ignored, travels with the gpurun snapshot); "base" is always built from the tree as it is.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/_variants
build_one() {   # name spec
  name=$1; spec=$2
  work=/tmp/zdr_variant_$name; rm -rf $work; mkdir -p $work
  cp -r $ROOT/zdr_amd $ROOT/include $work/
  flags="$spec"
  if [ "${spec#@}" != "$spec" ]; then (cd $work && patch -p1 -s < $ROOT/${spec#@}); flags=""; fi
  (cd $work && ZDR_KERNEL_FLAGS="$flags" python -c "import sys; sys.path.insert(0, '$work'); from zdr_amd import build; build.build(force=True)") > $work/build.log 2>&1 || { echo "variant $name FAILED:"; tail -5 $work/build.log; return 1; }
  cp $work/zdr_amd/csrc/libzdr_hip.so $ROOT/tools/_variants/libzdr_hip_$name.so
  echo "built $name [$spec]"
}
build_one base "" &
n=1
for kv in "$@"; do
  build_one "${kv%%=*}" "${kv#*=}" &
  n=$((n+1)); if [ $((n % 3)) = 0 ]; then wait; fi
done
wait
ls -la $ROOT/tools/_variants/
