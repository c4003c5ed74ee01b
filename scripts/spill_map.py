"""Where does a kernel spill?  python scripts/spill_map.py '<mangled-name-substring>' [-D flags]  -- lists every scratch access of
the fast build's kernel per basic block with its loop depth (no GPU needed)."""
import collections
import re
import subprocess
import sys

pat = sys.argv[1]
flags = sys.argv[2:]
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fvisibility=hidden", "-Iinclude", "--offload-arch=gfx950",
                       "-DTRG_STRICT=0"] + flags + ["-S", "--cuda-device-only", "toyraygun_amd/csrc/trg_kernels.hip", "-o", "/tmp/spill_map.s"],
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/spill_map.s").read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(")") is False and ":" in l][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
open("/tmp/spill_map_kernel.s", "w").write("\n".join(body))
ctx, out = "entry", collections.OrderedDict()
for l in body:
    if l.startswith(".LBB") or "; %bb." in l:
        ctx = re.sub(r"\s+", " ", l.strip())[:100]
    if "scratch_" in l:
        out.setdefault(ctx, []).append(l.strip().split(";")[0].strip())
for k, v in out.items():
    print(k, "\n     ", " | ".join(v))
print("instructions:", sum(1 for l in body if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")),
      " v_readlane/v_writelane:", sum(1 for l in body if "v_readlane" in l or "v_writelane" in l))
