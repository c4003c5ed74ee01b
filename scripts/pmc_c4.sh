#!/bin/bash
# memory-side counters of the C4 render kernel for one configuration (run on the GPU box):
#   bash scripts/pmc_c4.sh <tag> ["opt=value,opt=value"]      (options as TRG_EXP_OPTS, e.g. "12=0" = column tile order; TRG_HIP_SO=exp_build/<variant>/libtoyraygun_hip.so profiles a variant)
# one rocprofv3 --pmc pass per counter group of scripts/c4_run.py; prints per-launch medians of the render kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
tag=${1:-shipped}
export TRG_EXP_OPTS=$2
OUT=$PWD/gpurun_out/pmc_c4_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 scripts/c4_run.py 3 > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -3 "$OUT/$name.log"; }; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run ta TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE
python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] and "_kernel<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
out = dict(m)
if "FETCH_SIZE" in m: out["read_GB (2 x FETCH_SIZE KiB)"] = 2 * m["FETCH_SIZE"] * 1024 / 1e9
if "WRITE_SIZE" in m: out["write_GB"] = m["WRITE_SIZE"] * 1024 / 1e9
if "TCC_HIT_sum" in m: out["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
if "TCP_TCC_READ_REQ_sum" in m and "TCP_TOTAL_CACHE_ACCESSES_sum" in m: out["l1_miss_ratio"] = m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "TCP_TCC_READ_REQ_LATENCY_sum" in m: out["l1_to_l2_latency_cycles"] = m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"]
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m: out["lanes_per_valu"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]   # (of 64; round-4 verdict)
if "SQ_WAIT_ANY" in m: out["wait_share"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
for k in sorted(out): print("%-34s %.5g" % (k, out[k]))
json.dump(out, open(sys.argv[1] + "/../pmc_c4_%s.json" % sys.argv[2], "w"), indent=1)
PY
grep "C4" "$OUT/fetch.log" | tail -1
