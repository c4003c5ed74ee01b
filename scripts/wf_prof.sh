export TMPDIR=/tmp; export TRG_EXP_OPTS="5=2"
OUT=gpurun_out/wfpmc; rm -rf $OUT; mkdir -p $OUT
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 scripts/exp_ab.py --one=c4:shipped > $OUT/$name.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/wfpmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "wf_" in k or "render" in k: agg[k.split("::")[1][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k, {n:(len(v), round(sum(v)/len(v))) for n,v in c.items()})
PY
