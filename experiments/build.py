"""Builds experiments/lib/libtoyraygun_hip_exp.so: the product's HIP library + the schedules that were built, measured and lost
(the workgroup path pool, the wavefront schedule) -- the same sources with -DTRG_EXPERIMENTS=1.  The product library does not contain
them; tests/test_experiments.py runs one smoke test per experiment against this build (TRG_HIP_SO selects it for the ctypes binding).

  python experiments/build.py [--force]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
EXP_SO = os.path.join(HERE, "lib", "libtoyraygun_hip_exp.so")


W8_SO = os.path.join(HERE, "lib", "libtoyraygun_hip_w8.so")   # -DTRG_WIDE8=1: compressed 8-wide nodes for scenes in HBM (trg_wide8.inc.h)


def build(force=False, verbose=False):
    from toyraygun_amd import build as b
    b.build_hip_library(EXP_SO, obj_tag="exp_", defines=("-DTRG_EXPERIMENTS=1",), force=force, verbose=verbose)
    b.build_hip_library(W8_SO, obj_tag="w8_", defines=("-DTRG_WIDE8=1",), force=force, verbose=verbose, regen_shared=False)
    return EXP_SO


if __name__ == "__main__":
    print("built", build(force="--force" in sys.argv, verbose=True))
