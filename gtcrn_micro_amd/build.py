"""Builds the gfx950 shared library in-tree (gtcrn_micro_amd/libgtcrn_micro_hip.so).

hipcc cross-compiles for gfx950 without a GPU; the .so travels with the repo
snapshot to the GPU box.  No CPU fallback is built: the library is HIP only.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgtcrn_micro_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["kernels.hip", "api.cpp", "pack.cpp", "train_kernels.hip", "train.cpp"]
HEADERS = ["kernels.h", "layout.h", "pack.h", "train_kernels.h", os.path.join("..", "..", "include", "gtcrn_micro_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False, stamps=False, exp=False):
    """stamps=True builds the diagnostic variant libgtcrn_micro_hip_stamps.so (-DGT_STAMPS: in-kernel
    s_memtime phase stamps, used only by tools/phase_profile.py; never loaded by the product path).
    exp=True builds libgtcrn_micro_hip_exp.so with -DGT_EXP: whatever kernel experiment currently sits behind
    `#ifdef GT_EXP`, timed against the default build on the same GPU by tools/ab_bench.py."""
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]   # the flags live here
    objs = []
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
    # Inference kernels (kernels.hip):
    #  -fno-honor-nans: keeps hipcc from canonicalising (v_max x,x) in front of every v_min/v_max of the PReLU; no
    #     kernel tests for or produces NaN on finite input.
    #  -ffp-contract=on: multiply-adds are fused per source expression (by the front end), not opportunistically by the
    #     back end: every instantiation of a kernel template (one/two/three tiles per wave, offline / streaming /
    #     multi-stream) then rounds identically, which is what makes streamed == chunked == offline hold BIT FOR BIT.
    # Training kernels keep the defaults: NaN/Inf of a diverged run must propagate, and their fp32 statistics are
    # measurably more accurate with the back end's contraction (tests/test_gpu_train.py, fp64 comparison).
    per_file = {"kernels.hip": ["-fno-honor-nans", "-ffp-contract=on"]}
    suffix = ""
    lib = LIB
    if stamps:
        common.append("-DGT_STAMPS")
        per_file = {k: v + os.environ.get("GT_STAMPS_FLAGS", "").split() for k, v in per_file.items()}
        suffix = ".stamps"
        lib = LIB.replace(".so", "_stamps.so")
    elif exp:
        common.append("-DGT_EXP")
        per_file = {k: v + os.environ.get("GT_EXP_FLAGS", "").split() for k, v in per_file.items()}
        suffix = ".exp"
        lib = LIB.replace(".so", "_exp.so")
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + suffix + ".o")
        objs.append(obj)
        if force or _stale(obj, deps):
            cmd = [HIPCC, "--offload-arch=gfx950"] + common + per_file.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build_native(verbose=True))
