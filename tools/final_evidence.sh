# Collects what profiles/ and the docs quote for one round, from THIS tree, in two gpurun calls (a call is limited to 20 minutes):
#   bash tools/final_evidence.sh <tag> fd        AD against finite differences (≈14 min): copy gpurun_out/<tag>_fd_*.json into profiles/ BEFORE stage 2,
#                                                so that the bench line of stage 2 quotes them (they carry the hash of the sources they were measured on)
#   bash tools/final_evidence.sh <tag> profiles  counter passes c3 + c5, one kernel trace per configuration, the bench line, rehearsals (≈16 min)
cd $GRAFT_REPO_ROOT
tag=${1:-rX}; stage=${2:-profiles}; mkdir -p gpurun_out
if [ "$stage" = fd ]; then
  # the reference's fd_validate.py procedure (one pixel x one texel), diffuse and roughness, and the whole-image directional derivative
  timeout -k 10 200 python tools/fd_validate.py --channel diffuse --skip-table --tail-spp 4194304 --tail-seeds 1024 --out gpurun_out/${tag}_fd_validate_diffuse_texel.json > gpurun_out/${tag}_fd_validate_diffuse_texel.txt 2>&1 || echo "fd diffuse: FAILED or timed out"
  tail -2 gpurun_out/${tag}_fd_validate_diffuse_texel.txt
  timeout -k 10 180 python tools/fd_directional.py --seeds 256 --out gpurun_out/${tag}_fd_directional.json > gpurun_out/${tag}_fd_directional.txt 2>&1 || echo "fd directional: FAILED or timed out"
  tail -4 gpurun_out/${tag}_fd_directional.txt
  timeout -k 10 700 python tools/fd_validate.py --channel roughness --skip-table --tail-spp 4194304 --tail-seeds 7168 --out gpurun_out/${tag}_fd_validate_roughness_texel.json > gpurun_out/${tag}_fd_validate_roughness_texel.txt 2>&1 || echo "fd roughness: FAILED or timed out"
  tail -2 gpurun_out/${tag}_fd_validate_roughness_texel.txt
  exit 0
fi
bash tools/refresh_profiles.sh $tag > gpurun_out/${tag}_refresh.log 2>&1
grep "refresh_profiles\]" gpurun_out/${tag}_refresh.log
tail -c 1500 gpurun_out/${tag}_bench.json
timeout -k 10 100 python tools/trace_bench.py > gpurun_out/${tag}_trace_bench_1m.txt 2>&1; grep Mrays gpurun_out/${tag}_trace_bench_1m.txt
ZDR_DIST_BACKEND=gloo ZDR_SHARE_DEVICE=1 timeout -k 10 200 python bench.py --gpus 2 --steps 2 --warmup 1 2> gpurun_out/${tag}_two_rank_gloo.err | grep '^{' > gpurun_out/${tag}_two_rank_gloo_line.json
ls gpurun_out | grep "^${tag}_"
