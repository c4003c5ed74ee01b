"""Hash of what a render kernel's instruction counts depend on: ties imported profiler counters to the build they were measured on.

scripts/pmc_summary.py stores it in profiles/<round>/<config>_counters.json; bench.py recomputes it and refuses counters whose
hash differs from the tree it runs in (the instruction counts of a different kernel say nothing about this one)."""
import hashlib
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
ROOT = os.path.dirname(_HERE)


def kernel_source_files():
    """The kernels' own sources (trg_kernels.hip and everything it includes from csrc/), the ABI header they include
    (include/trg.h: the uniforms block, material and mask constants), and -- ADVICE r03 -- the host side that decides WHICH kernel
    runs with which parameters (trg_capi.cpp: kernel choice, frame split, regeneration lanes, tile order, LDS plan, scene layout):
    a changed schedule changes the per-launch instruction counts as surely as a changed kernel.  Since round 5 the BVH builders too."""
    names = ["trg_kernels.hip", "trg_kernels.h", "trg_device.h", "q4node.h"] + sorted(f for f in os.listdir(CSRC) if f.endswith(".inc.h")) + ["trg_capi.cpp"]
    # ... and the builders (round 5): the TREE a scene gets -- split rule, leaf sizes, quad pairing, the 4-wide collapse, on the host and on the
    # device (bench.py's c4xl leg builds on the device) -- sets the node steps per ray as surely as the traversal code does
    names += ["bvh_build.cpp", "bvh_build.h", "trg_build.hip"]
    return [os.path.join(CSRC, f) for f in names if os.path.exists(os.path.join(CSRC, f))] + [os.path.join(ROOT, "include", "trg.h")]


def _compile_flags():
    """The kernel compile flags of build.py (the `kern` / `common` lists and the -DTRG_* of the two kernel units), as source text."""
    src = open(os.path.join(_HERE, "build.py")).read()
    keep = [l.strip() for l in src.splitlines() if re.search(r"^\s*(common|hidden|dev|kern|regen_only)\s*=|trg_kernels_(fast|strict)(_regen)?\.o", l)]
    return "\n".join(keep)


BUILD_INFO = os.path.join(_HERE, "lib", "build_info.json")   # written by build.py next to the libraries it links


def compiler_version_from_hipcc():
    """`hipcc --version` of the compiler build.py would use, as "hip <version> clang <version>".  Raises when there is no hipcc."""
    from .build import _hipcc
    out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=120).stdout
    m = re.search(r"HIP version: *(\S+)", out)
    c = re.search(r"clang version *(\S+)", out)
    if not (m and c):
        raise RuntimeError("cannot read the compiler version from `hipcc --version`")
    return "hip %s clang %s" % (m.group(1), c.group(1))


def _compiler_version():
    """The compiler the libraries in lib/ were BUILT with (build.py records it in lib/build_info.json, which travels with them), or, for a
    tree that has not been built yet, the hipcc that would build it.  No silent stand-in: with neither, the hash cannot be formed and the
    caller is told so (round-4 verdict: a missing hipcc used to hash as the string "hipcc unavailable", every counters file then looked
    stale and bench.py quietly fell back to another roofline)."""
    try:
        import json
        with open(BUILD_INFO) as f:
            v = json.load(f).get("compiler")
        if v:
            return v
    except (OSError, ValueError):
        pass
    try:
        return compiler_version_from_hipcc()
    except Exception as e:
        raise RuntimeError("kernel_source_hash: neither %s nor a working hipcc tells which compiler built the kernels (%s: %s)"
                           % (os.path.relpath(BUILD_INFO, ROOT), type(e).__name__, e))


def kernel_source_hash():
    """sha256 over the names and contents of the files above, the kernel compile flags and the compiler version; first 16 hex digits."""
    h = hashlib.sha256()
    for path in kernel_source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(_compile_flags().encode() + b"\0")
    h.update(_compiler_version().encode() + b"\0")
    return h.hexdigest()[:16]
