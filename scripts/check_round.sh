# round-3 routine check on the GPU box: the GPU tests, the default bench line (with the C4 / C3 legs), the plain multi-GPU invocation
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3b/pytest.log
tail -4 gpurun_out/r3b/pytest.log
( time timeout -k 10 600 python bench.py ) > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err; echo "bench rc $?"
tail -3 gpurun_out/r3b/bench_default.err
timeout -k 10 120 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r3b/bench_g2.out 2>&1; echo "bench --gpus 2 rc $? (expected non-zero on a one-GPU box)"; tail -2 gpurun_out/r3b/bench_g2.out
# the same invocation rehearsed on this one GPU: 8 contexts on device 0, bands exchanged by peer copies (not a scaling number)
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 120 python bench.py --gpus 8 --no-cpu-baseline > gpurun_out/r3b/bench_g8_rehearsal.json 2> gpurun_out/r3b/bench_g8_rehearsal.err; echo "bench --gpus 8 rehearsal rc $?"
python - <<'PY'
import json
for l in open("gpurun_out/r3b/bench_default.json"):
    if l.startswith("{"):
        o = json.loads(l)
        print("C2 %.1f Mrays/s %.3f ms/step alone %.3f bound %s frac %s stale %s" % (o["value"], o["ms_per_step"], o["roofline"]["kernel_alone_ms"], o["roofline"]["bound"], o["roofline"]["frac"], o["roofline"].get("imported_stale")))
        for k, v in o.get("secondary", {}).items():
            print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_alone_ms", "error")}, v.get("roofline", {}).get("bound"), v.get("roofline", {}).get("frac"))
        print("cpu", o["cpu_baseline"]["value"], o["cpu_baseline"]["cores"])
PY
