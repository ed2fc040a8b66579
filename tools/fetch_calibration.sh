# Calibrates rocprofv3's FETCH_SIZE on a gather with a known byte count (MI355X_MICROARCH.md: "other access widths are
# uncalibrated: calibrate on a known byte count in your own access pattern"): tools/micro/gather_rate, 128 MiB of
# 64-byte records read at random, 4 x dwordx4 per lane-visit (every visit touches one distinct record).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_cal -- tools/micro/bin/gather_rate > gpurun_out/fetch_cal.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/fetch_cal/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    # dispatch order in gather_rate.hip: per size (1, 8, 128 MiB): modes 0..6, each a 20-visit warm-up then the 400-visit run
    big = rows[-14:]                       # the 128 MiB set
    names = ["per-lane 4 x dwordx4", "quad fetch + LDS", "per-lane 1 x dwordx4", "2 x dwordx4", "3 x dwordx4", "4 x dwordx2", "4 x dword"]
    for k, name in enumerate(names):
        kib = float(big[2 * k + 1]["Counter_Value"])
        visits = 256 * 20 * 400 * 64       # lane-visits of the timed launch
        print(f"{name:24s} FETCH_SIZE {kib:12.0f} KiB = {kib * 1024 / visits:6.1f} bytes per lane-visit (each visit reads 16-64 bytes of one random 64-byte record; 128 MiB set)")
PY
