# round 4, final kernels: the band balance (contiguous rows against interleaved micro-bands) and the kernel-only band sweep once more
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4bal2; mkdir -p $O
for c in c2 c4 c5; do
  timeout -k 10 300 python scripts/gpu_band_balance.py $c 2 4 8 2>/dev/null | grep "^{" > $O/balance_$c.jsonl; echo "$c rc $? $(wc -l < $O/balance_$c.jsonl) lines"
done
timeout -k 10 300 python scripts/gpu_overlap_bands.py > $O/overlap_bands.txt 2>&1; tail -12 $O/overlap_bands.txt
