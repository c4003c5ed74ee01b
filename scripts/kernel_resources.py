"""Register / scratch / occupancy of every kernel of trg_kernels.hip (no GPU needed):
python scripts/kernel_resources.py [extra -D flags]  -- what -Rpass-analysis=kernel-resource-usage reports for the shipped
(fast) build; `--strict` for the parity build."""
import re
import subprocess
import sys

flags = [a for a in sys.argv[1:] if a != "--strict"]
strict = "--strict" in sys.argv[1:]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fvisibility=hidden", "-Iinclude", "--offload-arch=gfx950",
       "-DTRG_STRICT=%d" % (1 if strict else 0)] + (["-ffp-contract=off"] if strict else []) + flags + \
      ["-Rpass-analysis=kernel-resource-usage", "-c", "toyraygun_amd/csrc/trg_kernels.hip", "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("trgk_fast::", "").replace("trgk_strict::", "").replace("void ", "")
    print("%-44s sgpr %3s (spilled %3s) vgpr %3s (spilled %3s) scratch %4s B/lane  occupancy %s" % (
        n, r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
