# round 4, final kernels (quad leaves, index / mask first in the plane records): GPU suite, counters of every configuration, C4 memory side per tile order
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4final3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > $O/gpu_tests.txt 2>&1; rc=$?
tail -3 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
TRG_COMMIT=$1 bash scripts/profile_round.sh c2 c4 c5 c3 c4xl > $O/profile.log 2>&1; rc=$?; tail -4 $O/profile.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
bash scripts/pmc_c4.sh order_columns "12=0" > $O/pmc_c4_columns.txt 2>&1; tail -6 $O/pmc_c4_columns.txt
bash scripts/pmc_c4.sh order_66_xcd_queues "12=66" > $O/pmc_c4_66.txt 2>&1; tail -6 $O/pmc_c4_66.txt
