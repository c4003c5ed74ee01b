#!/bin/bash
# FETCH_SIZE calibration (run on the GPU box through gpurun): bash scripts/fetch_calib.sh
# -> gpurun_out/calib/summary.txt: requested bytes per access shape next to what each memory-side read counter reports.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/calib; rm -rf "$OUT"; mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > "$OUT/requested.txt"
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- /tmp/fetch_calib > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -3 "$OUT/$name.log"; }
done
python3 - <<'PY'
import csv, glob, collections, re
out = "gpurun_out/calib"
req = {}
for l in open(f"{out}/requested.txt"):
    m = re.match(r"(\w+)\s+requested_bytes (\d+)", l)
    if m:
        req[m.group(1)] = int(m.group(2))
vals = collections.defaultdict(dict)
for f in glob.glob(f"{out}/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in req:
            if r["Kernel_Name"].startswith(k + "("):
                vals[k][r["Counter_Name"]] = float(r["Counter_Value"])
with open(f"{out}/summary.txt", "w") as fo:
    for k, b in req.items():
        v = vals[k]
        line = "%-14s requested %11d B" % (k, b)
        if "FETCH_SIZE" in v: line += " | FETCH_SIZE x1024 = %.3f x" % (v["FETCH_SIZE"] * 1024 / b)
        if "TCC_EA0_RDREQ_sum" in v: line += " | RDREQ x64 = %.3f x (32B reqs %d, BUBBLE %d)" % (v["TCC_EA0_RDREQ_sum"] * 64 / b, v.get("TCC_EA0_RDREQ_32B_sum", -1), v.get("TCC_BUBBLE_sum", -1))
        if "TCC_EA0_RDREQ_DRAM_32B_sum" in v: line += " | RDREQ_DRAM_32B x32 = %.3f x (RDREQ_DRAM %d)" % (v["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32 / b, v.get("TCC_EA0_RDREQ_DRAM_sum", -1))
        if "TCC_REQ_sum" in v: line += " | TCC hit %.3f" % (v.get("TCC_HIT_sum", 0) / max(v["TCC_REQ_sum"], 1))
        print(line); fo.write(line + "\n")
PY
