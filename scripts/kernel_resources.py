"""Register / scratch / occupancy of every kernel of trg_kernels.hip (no GPU needed):
python scripts/kernel_resources.py [extra -D flags]  -- what -Rpass-analysis=kernel-resource-usage reports for the shipped
(fast) build; `--strict` for the parity build."""
import re
import subprocess
import sys

flags = [a for a in sys.argv[1:] if a != "--strict"]
strict = "--strict" in sys.argv[1:]
# the shipped library builds trg_kernels.hip as two units (toyraygun_amd/build.py): the regeneration kernels apart (namespace trgk_regen_*), without
# the post-RA scheduler; only the kernels a unit LAUNCHES are listed for it
rows = []
for unit, extra in ((1, []), (2, ["-mllvm", "-enable-post-misched=0"])):
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fvisibility=hidden", "-Iinclude",
           "--offload-arch=gfx950", "-DTRG_STRICT=%d" % (1 if strict else 0), "-DTRG_UNIT=%d" % unit] + extra + (["-ffp-contract=off"] if strict else []) + flags + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", "toyraygun_amd/csrc/trg_kernels.hip", "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip(), "unit": unit}
            rows.append(cur)
        else:
            cur[k] = v
for r in rows:
    regen = "regen" in re.sub(r"\(.*", "", r["name"]).split("::")[-1]
    if (r["unit"] == 2) != regen:
        continue
    n = re.sub(r"\(.*", "", r["name"])
    n = re.sub(r"trgk\w*::", "", n).replace("void ", "")
    print("%-44s sgpr %3s (spilled %3s) vgpr %3s (spilled %3s) scratch %4s B/lane  occupancy %s" % (
        n, r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
