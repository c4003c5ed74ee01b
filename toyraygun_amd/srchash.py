"""Hash of the kernel sources: ties imported profiler counters to the build they were measured on.

scripts/pmc_summary.py stores it in profiles/<round>/<config>_counters.json; bench.py recomputes it and refuses counters whose
hash differs from the tree it runs in (the instruction counts of a different kernel say nothing about this one)."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def kernel_source_files():
    """What the render kernels are compiled from: trg_kernels.hip and everything it includes from this directory (the host-side
    sources -- C ABI, builders, device groups -- can change without making a kernel's instruction counts stale)."""
    names = ["trg_kernels.hip", "trg_kernels.h", "trg_device.h"] + sorted(f for f in os.listdir(CSRC) if f.endswith(".inc.h"))
    return [os.path.join(CSRC, f) for f in names]


def kernel_source_hash():
    """sha256 over the names and contents of the files above, first 16 hex digits."""
    h = hashlib.sha256()
    for path in kernel_source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]
