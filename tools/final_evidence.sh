# Collects what profiles/ and the docs quote for one round: bash tools/final_evidence.sh <tag>   (≈10 min of GPU time)
cd $GRAFT_REPO_ROOT
tag=${1:-rX}; out=gpurun_out/${tag}_final; mkdir -p $out
bash tools/refresh_profiles.sh $tag > $out/refresh.log 2>&1                       # c3: PMC passes, rocprof kernel stats, bench line
cp gpurun_out/${tag}_pmc.txt gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_bench.json $out/ 2>/dev/null
bash tools/pmc_big.sh $tag --spp 16 --iters 1 > $out/pmc_c5.txt 2>&1               # c5: lane utilisation, waits
timeout -k 10 400 python bench.py --config c5 --steps 3 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
# c5's kernel durations as rocprofv3 sees them (the program directly after `--`): 0.08 of the roofline must be recomputable from a kept csv
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_c5 -- python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $out/stats_c5.log 2>&1)
find gpurun_out/${tag}_stats_c5 -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_c5.csv \;
timeout -k 10 300 python tools/configs.py > $out/configs.txt 2>&1; cp gpurun_out/configs.json $out/configs.json
bash tools/rehearse_dist.sh > $out/two_rank_gloo_rehearsal.txt 2>&1
timeout -k 10 200 python tools/trace_bench.py > $out/trace_bench_1m.txt 2>&1
timeout -k 10 300 python tools/hotspot_bench.py --out $out/hotspot.json > $out/hotspot.txt 2>&1
timeout -k 10 600 python tests/measure/full_size_parity.py > $out/full_size_parity.txt 2>&1; cp gpurun_out/full_size_parity.json $out/ 2>/dev/null
ls $out
