# usage: bash tools/pmc_custom.sh <tag> "<counters of pass 1>;<counters of pass 2>;..." <run_pass args...>
# One rocprofv3 --pmc pass per ';'-separated group (counters only, no tracing), averaged per kernel.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; groups=$2; shift; shift
i=0
IFS=';' read -ra G <<< "$groups"
for pmc in "${G[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/pmcc_${tag}_$i -- python tools/run_pass.py "$@" > gpurun_out/pmcc_${tag}_$i.log 2>&1 || { echo "pass $i ($pmc) failed"; tail -3 gpurun_out/pmcc_${tag}_$i.log; }
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcc_${tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_path" not in k and "k_simple" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
