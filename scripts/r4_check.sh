# round 4 end-of-round rehearsal of what the driver runs: smoke(), the GPU tests, the default bench line, the launched 2-rank invocation over gloo on this one GPU
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4check; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
( time timeout -k 10 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -3 $O/bench_default.err
TRG_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_launched2_gloo.out 2>&1; echo "launched 2-rank (gloo rehearsal) rc $?"; tail -1 $O/bench_launched2_gloo.out | cut -c1-600
