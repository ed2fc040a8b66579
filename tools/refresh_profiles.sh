# Regenerates what bench.py and DESIGN.md quote from profiles/, on the GPU box, from THIS tree (every file is stamped with
# zdr_amd.build.source_hash() where bench.py checks it):
#   bash tools/refresh_profiles.sh <tag>    -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# One workload per profiler run: a kernel name then stands for one configuration and its AverageNs IS that configuration's figure
# (round 3 traced the default bench.py run, whose c3 and c4 legs launch the same kernel: only MinNs was usable).  The program goes
# directly after `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).  A pass that times out is SAID.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-rX}
note() { echo "[refresh_profiles] $*"; }

# ---- 1. counters, c3 (cbox path 512^2 spp 256) and c5 (1 M triangles, path 1024^2 spp 256): one rocprofv3 --pmc run per group
bash tools/pmc_passes.sh $tag --which both --iters 2 > gpurun_out/${tag}_pmc_c3.txt 2>&1 || note "PMC passes c3: FAILED or timed out"
bash tools/pmc_big.sh $tag --spp 256 --iters 1 > gpurun_out/${tag}_pmc_c5.txt 2>&1 || note "PMC passes c5: FAILED or timed out"
grep -l "failed" gpurun_out/${tag}_pmc_c3.txt gpurun_out/${tag}_pmc_c5.txt 2>/dev/null | while read f; do note "a counter pass failed: see $f"; done
python - <<PY
import re, json, ast, sys
sys.path.insert(0, ".")
from zdr_amd import build as hip_build
NOTE = ("rocprofv3 --pmc passes (one counter group per run), per launch. FETCH_SIZE / WRITE_SIZE are KiB. traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: "
        "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores and float atomics. "
        "For 16-byte gathers (texels, BVH nodes, records) the doubling is an upper bound (profiles/r2_fetch_size_calibration.txt). Infinity-Cache hits are counted: fabric traffic, an upper bound of HBM traffic.")
def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"void (k_path(?:_bwd)?)<.*?(\{.*\})\s*$", line)
        if not m or re.match(r"void k_path<\d, \w+, true", line): continue     # (the counting variant of zdr_render_stats is another kernel: STATS = true)
        rows.setdefault("k_path_bwd" if m.group(1) == "k_path_bwd" else "k_path_fwd", {}).update({k: float(v) for k, v in ast.literal_eval(m.group(2)).items()})
    return rows
def record(k, r):
    e = {"FETCH_SIZE_KiB": r.get("FETCH_SIZE"), "WRITE_SIZE_KiB": r.get("WRITE_SIZE"),
         "traffic_bytes": (2.0 * r.get("FETCH_SIZE", 0) + r.get("WRITE_SIZE", 0)) * 1024.0 if r.get("FETCH_SIZE") is not None else None,
         "SQ_INSTS_VALU": r.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": r.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVES": r.get("SQ_WAVES")}
    if r.get("SQ_THREAD_CYCLES_VALU") and r.get("SQ_ACTIVE_INST_VALU"): e["valu_lane_utilisation"] = r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_ACTIVE_INST_VALU"] * 64.0)
    if r.get("SQ_WAVE_CYCLES"): e["wait_any_frac"] = r.get("SQ_WAIT_ANY", 0) / r["SQ_WAVE_CYCLES"]; e["wait_inst_any_frac"] = r.get("SQ_WAIT_INST_ANY", 0) / r["SQ_WAVE_CYCLES"]
    if r.get("GRBM_GUI_ACTIVE"):
        e["gpu_cycles_per_xcd"] = r["GRBM_GUI_ACTIVE"] / 8.0
        if r.get("SQ_ACTIVE_INST_VALU"):   # VALU issue = SQ_ACTIVE_INST_VALU x 4 cycles per wave64 instruction on 1024 SIMDs
            e["valu_issue_frac"] = r["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * e["gpu_cycles_per_xcd"])
            e["measured_bound"] = "valu_issue" if e["valu_issue_frac"] > 0.6 else "mixed (see DESIGN.md 5)"
    if r.get("TCC_HIT_sum") is not None and r.get("TCC_MISS_sum") is not None and r["TCC_HIT_sum"] + r["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"])
    if k == "k_path_bwd" and r.get("TCC_EA0_ATOMIC_sum") is not None: e["atomic_requests"] = r["TCC_EA0_ATOMIC_sum"]
    return e
out = {"workload_key": "c3", "csrc_sha256": hip_build.source_hash(), "note": NOTE, "workload": "cbox path 512x512 spp 256 (tools/run_pass.py)"}
for k, r in parse("gpurun_out/${tag}_pmc_c3.txt").items(): out[k] = record(k, r)
c5 = {"workload": "1,004,672-triangle tessellated cbox, path 1024x1024 spp 256 (tools/run_big.py --spp 256: bench.py's c5 leg)"}
for k, r in parse("gpurun_out/${tag}_pmc_c5.txt").items(): c5[k] = record(k, r)
out["c5"] = c5
json.dump(out, open("gpurun_out/${tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json     # bench.py reads the traffic figures from here

# ---- 2. kernel durations as rocprofv3 sees them, ONE configuration per trace
trace() {   # trace <name> <seconds> <bench args...>
  name=$1; limit=$2; shift 2
  timeout -k 10 $limit rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$name -- python bench.py "$@" --no-extra-configs --no-cpu-baseline > gpurun_out/${tag}_stats_$name.log 2>&1
  rc=$?
  [ $rc -ne 0 ] && note "kernel trace $name: rocprofv3 ended with $rc (124 = timed out after $limit s) — its csv is INCOMPLETE or missing"
  find gpurun_out/${tag}_stats_$name -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats_$name.csv \;
  [ -f gpurun_out/${tag}_kernel_stats_$name.csv ] && head -4 gpurun_out/${tag}_kernel_stats_$name.csv
}
trace c3 300 --steps 10 --warmup 2
trace c2 200 --config c2 --steps 10 --warmup 2
trace c4_one_gpu 300 --config c4 --steps 3 --warmup 1
trace c5 400 --config c5 --steps 3 --warmup 1

# ---- 3. the bench line itself (default run: headline + c2 / c4 on one GPU / c5 legs + CPU baseline)
timeout -k 10 600 python bench.py --steps 10 --warmup 2 2>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json || note "bench.py: FAILED or timed out"
cat gpurun_out/${tag}_bench.json
