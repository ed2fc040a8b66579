# Is the vector-memory path (TA / TCP) a limit of the 1 M-triangle kernels?  Busy / stall counters of the texture-address and L1 units.
#   bash tools/pmc_ta.sh <tag> <run_big args...>     (one counter group per run; a group with an unknown name just fails)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 -L > gpurun_out/pmc_ta_${tag}_list.txt 2>&1
i=0
for pmc in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/pmcta_${tag}_$i -- python tools/run_big.py "$@" > gpurun_out/pmcta_${tag}_$i.log 2>&1 || echo "pass $i failed: $pmc"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcta_${tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:44]
            if "k_path" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
