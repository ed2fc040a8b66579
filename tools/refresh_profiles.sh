# Regenerates everything under profiles/ that bench.py and DESIGN.md quote, on the GPU box:
#   bash tools/refresh_profiles.sh <tag>    -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc.txt, <tag>_pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-rX}
bash tools/pmc_passes.sh $tag --which both --iters 2 > gpurun_out/${tag}_pmc.txt 2>&1 || true
python - <<PY
import re, json, ast, sys
sys.path.insert(0, ".")
from zdr_amd import build as hip_build
rows = {}
for line in open("gpurun_out/${tag}_pmc.txt"):
    m = re.match(r"void (k_path(?:_bwd)?)<.*?(\{.*\})\s*$", line)
    if not m: continue
    rows.setdefault("k_path_bwd" if m.group(1) == "k_path_bwd" else "k_path_fwd", {}).update({k: float(v) for k, v in ast.literal_eval(m.group(2)).items()})
out = {"workload_key": "c3", "csrc_sha256": hip_build.source_hash(), "note": "rocprofv3 --pmc passes (tools/pmc_passes.sh, one counter group per run), cbox path 512x512 spp256, per launch. FETCH_SIZE / WRITE_SIZE are KiB. traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores and float atomics. The kernel's reads are 16-byte-per-lane streams (FIFO entries) and 16-byte gathers (texels, records); for the gathers the doubling is an upper bound (profiles/r2_fetch_size_calibration.txt). Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of HBM traffic."}
for k, r in rows.items():
    e = {"FETCH_SIZE_KiB": r.get("FETCH_SIZE"), "WRITE_SIZE_KiB": r.get("WRITE_SIZE"),
         "traffic_bytes": (2.0 * r.get("FETCH_SIZE", 0) + r.get("WRITE_SIZE", 0)) * 1024.0,
         "valu_lane_utilisation": r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_ACTIVE_INST_VALU"] * 64.0),
         "SQ_INSTS_VALU": r["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": r["SQ_ACTIVE_INST_VALU"], "SQ_WAVES": r["SQ_WAVES"],
         "wait_any_frac": r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], "wait_inst_any_frac": r["SQ_WAIT_INST_ANY"] / r["SQ_WAVE_CYCLES"],
         "gpu_cycles_per_xcd": r["GRBM_GUI_ACTIVE"] / 8.0}
    if k == "k_path_bwd": e["atomic_requests"] = r.get("TCC_EA0_ATOMIC_sum")
    # what the counters say bounds the kernel: VALU issue = SQ_ACTIVE_INST_VALU x 4 cycles per wave64 instruction on 1024 SIMDs
    e["valu_issue_frac"] = r["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * e["gpu_cycles_per_xcd"])
    e["measured_bound"] = "valu_issue" if e["valu_issue_frac"] > 0.6 else "mixed (see DESIGN.md 5)"
    out[k] = e
json.dump(out, open("gpurun_out/${tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json     # bench.py reads the traffic figure from here
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1 || true
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
timeout -k 10 600 python bench.py --steps 10 --warmup 2 2>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_bench.json; head -5 gpurun_out/${tag}_kernel_stats.csv
