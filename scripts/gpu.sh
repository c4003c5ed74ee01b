#!/bin/bash
# The ONE parameterised runner for work on the GPU box (replaces the per-run r4_*.sh command lists, check_round.sh, ab.sh):
#
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
#
# Output goes to gpurun_out/<tag>/ (merged back by gpurun); the steps run in order and the run STOPS at the first one that fails (no GPU
# step is started after a failed or timed-out one).  Steps:
#   tests[:<expr>]        pytest -m gpu [-k <expr>]                       -> tests.log
#   slow                  pytest -m "gpu and slow" (whole-frame parity)   -> tests_slow.log
#   exptests              the whole -m gpu suite against experiments/lib/libtoyraygun_hip_exp.so (all three schedules)
#   smoke                 __graft_entry__.smoke()                          -> smoke.log
#   bench[:<cfg>]         python bench.py [--config cfg]                   -> bench_<cfg>.json (+ the detail file)
#   dbench:<cfg>          the same with the device binned-SAH builder
#   g<N>                  bench.py --gpus N rehearsed on this one device (N contexts, copy exchange)  -> bench_g<N>.json
#   group1                N = 1 through the group path (TRG_BENCH_GROUP=1)
#   profile:<cfg,...>     scripts/profile_round.sh for those configurations -> gpurun_out/profiles/
#   pmc:<script>:<tag>[:<opts>]   bash scripts/<script>.sh <tag> <opts>  (pmc_c4, pmc_c2, pmc_c3_tail ...)
#   ab:<cfg>:<v1,v2,...>[:<reps>]   scripts/exp_ab.py --one=<cfg>:<v> for every variant (exp_build.sh names, or "shipped"), <reps> rounds (2)
#   env:<NAME>=<value>    export for the steps that follow (e.g. env:TRG_EXP_OPTS=12=66)
#   fuzz:<n>:<seed>[:fast]  scripts/gpu_fuzz.py
#   py:<script>[:<arg>...]  python scripts/<script>.py <args>             -> <script>.log
#   resources             scripts/kernel_resources.py                      -> kernel_resources.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag; mkdir -p "$O"
step() { echo "== $*"; }
for s in "$@"; do
  IFS=: read -r what a b c <<< "$s"
  case "$what" in
    tests)   step "pytest -m gpu ${a:+-k $a}"
             if [ -n "$a" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$a" > "$O/tests.log" 2>&1; else timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; fi
             rc=$?; tail -4 "$O/tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" "$O/tests.log" | tail -20; exit $rc; } ;;
    slow)    step "pytest -m 'gpu and slow'"; timeout -k 10 1100 python -m pytest tests -m "gpu and slow" -x -q > "$O/tests_slow.log" 2>&1; rc=$?; tail -4 "$O/tests_slow.log"; [ $rc -eq 0 ] || exit $rc ;;
    exptests) step "pytest -m gpu on the experiments build"
             TRG_HIP_SO=$PWD/experiments/lib/libtoyraygun_hip_exp.so timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$O/tests_exp.log" 2>&1; rc=$?; tail -4 "$O/tests_exp.log"; [ $rc -eq 0 ] || exit $rc ;;
    smoke)   step smoke; timeout -k 10 300 python __graft_entry__.py --smoke > "$O/smoke.log" 2>&1; rc=$?; tail -1 "$O/smoke.log"; [ $rc -eq 0 ] || exit $rc ;;
    bench|dbench)
             cfg=${a:-c2}; step "bench $cfg"
             extra=""; [ "$cfg" != c2 ] && extra="--config $cfg --no-cpu-baseline"
             [ "$what" = dbench ] && export TRG_BENCH_GPU_BUILD=1
             ( time TRG_BENCH_DETAIL=$PWD/$O/bench_${what}_${cfg}_detail.json timeout -k 10 600 python bench.py $extra ) > "$O/bench_${what}_${cfg}.json" 2> "$O/bench_${what}_${cfg}.err"; rc=$?
             unset TRG_BENCH_GPU_BUILD
             tail -4 "$O/bench_${what}_${cfg}.err"; [ $rc -eq 0 ] || exit $rc
             python - "$O/bench_${what}_${cfg}.json" <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r, c = o["roofline"], o["config"]
print("%s %.1f Mrays/s %.4f ms/step alone %.3f bound %s frac %s lanes %s stale %s line %d B" % (c["name"], o["value"], o["ms_per_step"], r["kernel_alone_ms"], r["bound"], r["frac"], r.get("lanes"), r.get("imported_stale"), len(json.dumps(o))))
for leg in ("c4", "c3", "c4xl"):
    if leg + "_mrays" in c:
        print(" ", leg, {k[len(leg) + 1:]: v for k, v in c.items() if k.startswith(leg + "_") and k.split("_", 1)[1] in ("mrays", "ms_per_step", "alone_ms", "frac", "valu_frac", "lanes", "hbm_frac", "l2_hit", "error")})
if "plugin_fps" in c:
    print("  plugin", {k[7:]: v for k, v in c.items() if k.startswith("plugin_")})
PY
             ;;
    g[0-9]*) n=${what#g}; step "bench --gpus $n on one device"
             devs=$(python -c "print(','.join(['0']*$n))")
             TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=$devs TRG_BENCH_DETAIL=$PWD/$O/bench_g${n}_detail.json timeout -k 10 300 python bench.py --gpus $n --no-cpu-baseline ${a:+--config $a} > "$O/bench_g$n.json" 2> "$O/bench_g$n.err"; rc=$?
             [ $rc -eq 0 ] || { tail -5 "$O/bench_g$n.err"; exit $rc; } ;;
    group1)  step "N = 1 through the group path"; TRG_BENCH_GROUP=1 TRG_BENCH_DETAIL=$PWD/$O/bench_group1_detail.json timeout -k 10 300 python bench.py --no-cpu-baseline > "$O/bench_group1.json" 2> "$O/bench_group1.err"; rc=$?; [ $rc -eq 0 ] || exit $rc ;;
    profile) step "profile_round ${a//,/ }"; TRG_COMMIT=$(cat .commit 2>/dev/null) timeout -k 10 1100 bash scripts/profile_round.sh ${a//,/ } > "$O/profile_round.log" 2>&1; rc=$?; tail -12 "$O/profile_round.log"; [ $rc -eq 0 ] || exit $rc ;;
    pmc)     step "$a $b $c"; timeout -k 10 900 bash "scripts/$a.sh" "$b" "$c" > "$O/$a_$b.log" 2>&1; rc=$?; tail -15 "$O/$a_$b.log"; [ $rc -eq 0 ] || exit $rc ;;
    ab)      reps=${c:-2}
             for rep in $(seq 1 "$reps"); do for v in ${b//,/ }; do
               timeout -k 10 240 python scripts/exp_ab.py --one="$a:$v" 2>&1 | grep -v amdgpu.ids | tee -a "$O/ab_$a.log"; [ "${PIPESTATUS[0]}" -eq 0 ] || exit 1
             done; done ;;
    env)     export "$a${b:+:$b}${c:+:$c}"; echo "export $a${b:+:$b}${c:+:$c}" ;;
    fuzz)    step "fuzz $a cases, seed $b $c"; timeout -k 10 1000 python scripts/gpu_fuzz.py "$a" "$b" $c > "$O/fuzz_${c:-strict}_seed$b.log" 2>&1; rc=$?; tail -3 "$O/fuzz_${c:-strict}_seed$b.log"; [ $rc -eq 0 ] || exit $rc ;;
    py)      step "python scripts/$a.py $b $c"; timeout -k 10 1000 python "scripts/$a.py" $b $c > "$O/$a${b:+_$b}.log" 2>&1; rc=$?; tail -25 "$O/$a${b:+_$b}.log"; [ $rc -eq 0 ] || exit $rc ;;
    resources) python scripts/kernel_resources.py > "$O/kernel_resources.txt" 2>&1; tail -3 "$O/kernel_resources.txt" ;;
    *)       echo "unknown step: $s"; exit 2 ;;
  esac
done
echo "gpu.sh $tag: all steps done"
