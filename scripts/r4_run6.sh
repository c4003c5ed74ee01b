O=gpurun_out/r4f; mkdir -p $O
bash scripts/ab.sh c2 shipped planes0 2>&1 | tail -4
bash scripts/ab.sh c3 shipped planes0 2>&1 | tail -4
bash scripts/pmc_c2.sh planes1 2>&1 | tail -12
bash scripts/pmc_c2.sh planes0 planes0 2>&1 | tail -12
