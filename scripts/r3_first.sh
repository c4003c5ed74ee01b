set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
for o in 0 2 1 4 8; do
  echo "== tile order $o" | tee -a gpurun_out/r3a/ab.log
  TRG_EXP_OPTS="12=$o" timeout -k 10 120 python scripts/exp_ab.py --one=c4:shipped 2>&1 | tee -a gpurun_out/r3a/ab.log
done
timeout -k 10 120 python scripts/exp_ab.py --one=c2:shipped 2>&1 | tee -a gpurun_out/r3a/ab.log
timeout -k 10 120 python scripts/exp_ab.py --one=c3:shipped 2>&1 | tee -a gpurun_out/r3a/ab.log
