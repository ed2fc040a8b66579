# How full are the trips, the sweep loop and the flushes of the cbox backward kernel?  A MEASUREMENT build (-DZDR_MEASURE_STATS: counters in
# k_path_bwd and scatter_flush, printed by zdr_render_backward) over the bench workload; the shipped library is rebuilt afterwards.
#   bash tools/bwd_stats.sh [run_pass args]   -> stdout
cd $GRAFT_REPO_ROOT
ZDR_KERNEL_FLAGS="-DZDR_MEASURE_STATS" python -m zdr_amd.build --force > /dev/null 2>gpurun_out/bwd_stats_build.log || { tail -5 gpurun_out/bwd_stats_build.log; exit 1; }
timeout -k 10 200 python tools/run_pass.py --which bwd --iters 1 "$@" 2>&1 | grep -E "bwd stats|bwd:" | tail -2 | python -c "
import sys, re
for line in sys.stdin:
    print(line.rstrip())
    m = dict(re.findall(r'(\w+) (\d+)', line))
    if 'trips' in m:
        t, sh, fin, it, st, fl, en, du = (float(m[k]) for k in ('trips', 'shaded', 'finished', 'sweep_iterations', 'sweep_steps', 'flushes', 'entries', 'duplicate_cells'))
        print(f'  per trip: {sh / t:.1f} vertices shaded, {fin / t:.1f} paths end, sweep loop {it / t:.2f} iterations carrying {st / t:.1f} steps = {st / (64 * it):.3f} of the lanes per iteration')
        print(f'  flushes: {en / fl:.1f} entries each, {du / en:.4f} of the entries repeat a cell of the same flush ({en / sh:.3f} entries per shaded vertex)')
"
python -m zdr_amd.build --force > /dev/null 2>&1
