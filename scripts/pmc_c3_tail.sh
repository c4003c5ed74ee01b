#!/bin/bash
# counters of the C3 tail kernel with / without TRG_OPT_TAIL_SORT (run on the GPU box): bash scripts/pmc_c3_tail.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_c3_tail; rm -rf "$OUT"; mkdir -p "$OUT"
for mode in 0 1 3; do
  export TRG_EXP_OPTS="14=$mode"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/m$mode" -- python3 scripts/exp_ab.py --one=c3:shipped > "$OUT/m$mode.log" 2>&1 || tail -3 "$OUT/m$mode.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
res = {}
for mode in (0, 1, 3):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(sys.argv[1] + "/m%d/*/*_counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for k, c in agg.items():
        if "tail_kernel" in k or "tail_sort" in k or "head_kernel" in k:
            out[k] = {"valu_insts": c["SQ_INSTS_VALU"], "lanes_per_valu": c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_INSTS_VALU"], 1), "lds_insts": c["SQ_INSTS_LDS"],
                      "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"], "lds_idx_active_cycles": c["SQ_LDS_IDX_ACTIVE"],
                      "bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), "wait_share": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1)}
    res["sort_mode_%d" % mode] = out
    print(mode, json.dumps(out))
json.dump(res, open(sys.argv[1] + "/../pmc_c3_tail.json", "w"), indent=1)
PY
