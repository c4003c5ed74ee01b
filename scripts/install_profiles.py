"""Copies what scripts/profile_round.sh left under gpurun_out/profiles/ into profiles/<round>/ (python scripts/install_profiles.py r03):
the per-config counters files (stamped with the commit they were taken at: the GPU box has no .git), the kernel-stats tables, the full
summary and the kernel resources of the shipped build.  bench.py then accepts the counters as long as the kernel sources do not change."""
import json, os, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = "gpurun_out/profiles", os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
sys.path.insert(0, ".")
from toyraygun_amd.srchash import kernel_source_hash
here = kernel_source_hash()
for cfg in ("c2", "c3", "c4", "c5", "c4xl"):
    p = os.path.join(src, cfg + "_counters.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    if d.get("kernel_source_hash") != here:
        sys.exit("%s was taken on other kernel sources (%s, this tree %s): profile again" % (p, d.get("kernel_source_hash"), here))
    d["commit"] = d.get("commit") or commit
    d["source"] = d.get("source", "").replace("at commit :", "at commit %s:" % commit)
    json.dump(d, open(os.path.join(dst, cfg + "_counters.json"), "w"), indent=1)
    shutil.copy(os.path.join(src, cfg + "_kernel_stats.csv"), dst)
    print(cfg, d["kernel_source_hash"], "VALU %.4g" % d["valu_insts_per_launch"], "lanes %.2f" % (d.get("lanes_active_per_valu_inst") or 0))
shutil.copy(os.path.join(src, "summary.json"), dst)
open(os.path.join(dst, "kernel_resources.txt"), "w").write(subprocess.run([sys.executable, "scripts/kernel_resources.py"], capture_output=True, text=True).stdout)
