# PMC passes over the 1M-triangle path workload (tools/run_big.py): bash tools/pmc_big.sh <tag> <run_big args...>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/pmcbig_${tag}_$i -- python tools/run_big.py "$@" > gpurun_out/pmcbig_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcbig_${tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_path" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
