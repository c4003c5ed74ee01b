"""Builds the in-tree native libraries with hipcc (gfx950 only):

  toyraygun_amd/lib/libtoyraygun_hip.so   HIP kernels + the C ABI of include/trg.h
  toyraygun_amd/lib/libtoyraygun.so       host C++ plugin surface (Engine/Renderer/Scene/HipRenderer)
  toyraygun_amd/lib/toyraygun_cornell     demo app following the reference main.cpp call order

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(CSRC, "host")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
ROOT = os.path.dirname(HERE)
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _glob(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts)) if os.path.isdir(d) else []


def build_hip_library(hip_so, obj_tag="", defines=(), force=False, verbose=False, regen_shared=True):
    """Compile the HIP side (kernels in two builds x two translation units, the device builders, the C ABI, the host BVH builder, the device
    group) and link it into `hip_so`.  `defines` / `obj_tag`: a variant of the same sources (experiments/build.py: -DTRG_EXPERIMENTS=1)."""
    os.makedirs(os.path.dirname(hip_so), exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    exp_dir = os.path.join(ROOT, "experiments")
    headers = _glob(CSRC, (".h",)) + _glob(os.path.join(ROOT, "include"), (".h",)) + \
        _glob(os.path.join(ROOT, "include", "engine"), (".h",)) + _glob(os.path.join(ROOT, "include", "bx"), (".h",)) + \
        _glob(HOST, (".h",)) + [os.path.abspath(__file__)] + (_glob(exp_dir, (".h",)) if defines else [])
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + list(defines)
    hidden = ["-fvisibility=hidden"]
    dev = ["--offload-arch=" + ARCH]
    # -fno-slp-vectorize: the SLP vectoriser pairs the Halton digit chains of two dimensions into v_pk_mul_f32 /
    # v_pk_fma_f32, which are not faster than two scalar ops on gfx950 and cost 10 VGPRs + scratch spills
    # (80 VGPRs + 48 B scratch -> 70 VGPRs, none): +9 % on C2, +2.5 % on C4 (NOTEBOOK.md, 'registers')
    # -mllvm -amdgpu-sched-strategy=max-ilp (round 4): the instruction scheduler's ILP-first strategy; same instructions, other order --
    # render_kernel<LDS scene> 61 VGPRs + 12 in scratch -> 63 and none, C2 -1.2 %, C3 -1.0 % time, C4 unchanged (profiles/r04/ab_sched_strategy.txt)
    kern = dev + ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    regen_only = ["-mllvm", "-enable-post-misched=0"]

    objs = []
    units = [
        # trg_kernels.hip as two translation units per build: 1 = everything but the path-regeneration kernels, 2 = those alone, without the
        # post-RA scheduler (C4 -1.8 % time; the other kernels lose up to 0.4 % to that flag: profiles/r04/ab_sched_strategy.txt)
        ("trg_kernels_fast.o", os.path.join(CSRC, "trg_kernels.hip"), kern + ["-DTRG_STRICT=0", "-DTRG_UNIT=1"]),
        ("trg_kernels_fast_regen.o", os.path.join(CSRC, "trg_kernels.hip"), kern + regen_only + ["-DTRG_STRICT=0", "-DTRG_UNIT=2"]),
        ("trg_kernels_strict.o", os.path.join(CSRC, "trg_kernels.hip"), kern + ["-DTRG_STRICT=1", "-ffp-contract=off", "-DTRG_UNIT=1"]),
        ("trg_kernels_strict_regen.o", os.path.join(CSRC, "trg_kernels.hip"), kern + regen_only + ["-DTRG_STRICT=1", "-ffp-contract=off", "-DTRG_UNIT=2"]),
        ("trg_build.o", os.path.join(CSRC, "trg_build.hip"), dev),
        ("trg_capi.o", os.path.join(CSRC, "trg_capi.cpp"), ["-x", "hip", "--offload-arch=" + ARCH]),
        ("bvh_build.o", os.path.join(CSRC, "bvh_build.cpp"), ["-x", "hip", "--offload-arch=" + ARCH]),
        ("trg_group.o", os.path.join(CSRC, "trg_group.cpp"), ["-x", "hip", "--offload-arch=" + ARCH, "-I/opt/rocm/include"]),
    ]
    # (the regeneration units do not contain the experimental schedules: the variant shares those objects with the product build)
    shared_with_product = (("trg_kernels_fast_regen.o", "trg_kernels_strict_regen.o") if regen_shared else ()) + ("trg_build.o", "bvh_build.o", "trg_group.o")
    for name, src, extra in units:
        o = os.path.join(OBJ, name if (not obj_tag or name in shared_with_product) else obj_tag + name)
        if force or _newer(o, [src] + headers):
            _run([hipcc] + common + hidden + extra + ["-c", src, "-o", o], verbose)
        objs.append(o)
    if force or _newer(hip_so, objs):
        _run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", hip_so] + objs + ["-ldl", "-lpthread"], verbose)
        return True
    return False


def build(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    hip_so = os.path.join(LIB, "libtoyraygun_hip.so")
    info = os.path.join(LIB, "build_info.json")
    headers = _glob(CSRC, (".h",)) + _glob(os.path.join(ROOT, "include"), (".h",)) + \
        _glob(os.path.join(ROOT, "include", "engine"), (".h",)) + _glob(os.path.join(ROOT, "include", "bx"), (".h",)) + \
        _glob(HOST, (".h",)) + [os.path.abspath(__file__)]
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]
    linked = build_hip_library(hip_so, force=force, verbose=verbose)
    if linked or not os.path.exists(info):
        # which compiler built what sits in lib/: srchash.py ties imported profiler counters to it (the file travels with the libraries)
        import json
        from .srchash import compiler_version_from_hipcc
        with open(info, "w") as f:
            json.dump({"compiler": compiler_version_from_hipcc(), "hipcc": _hipcc()}, f)

    # host C++ plugin surface (pure host code; links against the C ABI only)
    host_srcs = _glob(HOST, (".cpp",))
    host_so = os.path.join(LIB, "libtoyraygun.so")
    if host_srcs:
        hobjs = []
        for src in host_srcs:
            if os.path.basename(src) == "main_cornell.cpp":
                continue
            o = os.path.join(OBJ, "host_" + os.path.basename(src)[:-4] + ".o")
            if force or _newer(o, [src] + headers):
                _run(["g++"] + common + ["-fno-exceptions", "-fno-rtti", "-c", src, "-o", o], verbose)
            hobjs.append(o)
        if force or _newer(host_so, hobjs + [hip_so]):
            _run(["g++", "-shared", "-fPIC", "-o", host_so] + hobjs +
                 ["-L" + LIB, "-ltoyraygun_hip", "-Wl,-rpath,$ORIGIN"], verbose)
        app_src = os.path.join(HOST, "main_cornell.cpp")
        app = os.path.join(LIB, "toyraygun_cornell")
        if os.path.exists(app_src) and (force or _newer(app, [app_src, host_so] + headers)):
            _run(["g++"] + common + ["-fno-exceptions", "-fno-rtti", app_src, "-o", app, "-L" + LIB,
                                     "-ltoyraygun", "-ltoyraygun_hip", "-Wl,-rpath,$ORIGIN"], verbose)
    return hip_so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
