for ml in 1 2 4; do echo "maxleaf=$ml"; TRG_BVH_MAXLEAF=$ml timeout -k 10 120 python scripts/exp_time.py base 2>&1 | grep lds=; done
timeout -k 10 300 python scripts/exp_time.py w6 w8 2>&1 | grep lds=
