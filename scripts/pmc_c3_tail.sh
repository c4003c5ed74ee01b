#!/bin/bash
# Counters and durations of the C3 head / tail kernels for a list of option sets (run on the GPU box):
#   bash scripts/pmc_c3_tail.sh <tag> "<TRG_EXP_OPTS of variant 1>" "<TRG_EXP_OPTS of variant 2>" ...      e.g.  "15=0" "15=1"  (tail refill off / on)
# per variant one --kernel-trace --stats pass and three --pmc passes of scripts/exp_ab.py --one=c3:<lib> (C3 at 64 spp; TRG_HIP_LIB_VARIANT names an
# exp_build variant, default shipped).  Output: gpurun_out/pmc_c3_tail_<tag>.json
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
tag=$1; shift
lib=${TRG_HIP_LIB_VARIANT:-shipped}
OUT=$PWD/gpurun_out/pmc_c3_tail_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for opts in "$@"; do
  export TRG_EXP_OPTS="$opts"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/v$i/trace" -- python3 scripts/exp_ab.py --one=c3:$lib > "$OUT/v$i.trace.log" 2>&1 || tail -3 "$OUT/v$i.trace.log"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/v$i/sq" -- python3 scripts/exp_ab.py --one=c3:$lib > "$OUT/v$i.sq.log" 2>&1 || tail -3 "$OUT/v$i.sq.log"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/v$i/lds" -- python3 scripts/exp_ab.py --one=c3:$lib > "$OUT/v$i.lds.log" 2>&1 || tail -3 "$OUT/v$i.lds.log"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/v$i/fetch" -- python3 scripts/exp_ab.py --one=c3:$lib > "$OUT/v$i.fetch.log" 2>&1 || tail -3 "$OUT/v$i.fetch.log"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/v$i/write" -- python3 scripts/exp_ab.py --one=c3:$lib > "$OUT/v$i.write.log" 2>&1 || tail -3 "$OUT/v$i.write.log"
  grep "pipelined" "$OUT/v$i.trace.log" | tail -1
  i=$((i+1))
done
python3 - "$OUT" "$tag" "$@" <<'PY'
import csv, glob, sys, collections, json
out_dir, tag, variants = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for i, opts in enumerate(variants):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("%s/v%d/*/*/*_counter_collection.csv" % (out_dir, i)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("%s/v%d/trace/*/*_kernel_trace.csv" % (out_dir, i)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    tot = {"valu_insts": 0.0, "thread_cycles": 0.0, "active_inst": 0.0, "fetch": 0.0, "write": 0.0, "ms": 0.0}
    for k, c in agg.items():
        if not any(s in k for s in ("tail_kernel", "tail_sort", "head_kernel", "tail_accumulate")):
            continue
        out[k] = {"launches": dur[k][0], "total_ms": round(dur[k][1], 3), "valu_insts": c["SQ_INSTS_VALU"],
                  "lanes_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1), "salu_insts": c["SQ_INSTS_SALU"], "lds_insts": c["SQ_INSTS_LDS"],
                  "vmem_rd_insts": c["SQ_INSTS_VMEM_RD"], "vmem_wr_insts": c["SQ_INSTS_VMEM_WR"],
                  "bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), "wait_mem_share": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                  "wait_issue_share": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                  "read_GB": 2 * c["FETCH_SIZE"] * 1024 / 1e9, "write_GB": c["WRITE_SIZE"] * 1024 / 1e9}
        tot["valu_insts"] += c["SQ_INSTS_VALU"]; tot["thread_cycles"] += c["SQ_THREAD_CYCLES_VALU"]; tot["active_inst"] += c["SQ_ACTIVE_INST_VALU"]
        tot["fetch"] += 2 * c["FETCH_SIZE"] * 1024 / 1e9; tot["write"] += c["WRITE_SIZE"] * 1024 / 1e9; tot["ms"] += dur[k][1]
    out["_all_tail_path_kernels"] = {"valu_insts": tot["valu_insts"], "lanes_per_valu_inst": tot["thread_cycles"] / max(tot["active_inst"], 1), "read_GB": tot["fetch"], "write_GB": tot["write"], "kernel_ms_sum": round(tot["ms"], 2),
                                     "note": "sums over every launch of the profiled process (exp_ab.py: 1 counted + 5 alone + 6 + K pipelined renders of C3 at 64 spp)"}
    res[opts] = out
    print("==", opts)
    for k, v in out.items():
        print("  %-28s %s" % (k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "note"}))
json.dump(res, open("%s/../pmc_c3_tail_%s.json" % (out_dir, tag), "w"), indent=1)
PY
