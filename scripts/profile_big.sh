#!/bin/bash
# HBM traffic of the megakernel on the 10.6 M-triangle lattice (working set beyond L2 + Infinity Cache)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_big; rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 scripts/gpu_big.py 96"
run() { name=$1; shift; echo "pass $name"; timeout -k 10 300 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -3 "$OUT/$name.log"; }; }
run trace --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD
run fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD
run write --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- $CMD
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_big"
for f in sorted(glob.glob(f"{out}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"] and ", true>" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[2], {k: (len(v), round(sum(v) / len(v))) for k, v in agg.items()})
for f in glob.glob(f"{out}/trace/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:3]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
