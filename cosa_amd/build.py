"""Build libcosa_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m cosa_amd.build [--force]

Each .hip file is compiled to an object with its own flags and linked into
cosa_amd/lib/libcosa_hip.so.  The "exact" translation units (label / PAR / lattice) are built
with -ffp-contract=off so that they follow the arithmetic spec literally.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libcosa_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-fPIC", "--offload-arch=" + ARCH, "-std=c++17", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(HERE, "..", "include")]
EXACT = ["-ffp-contract=off", "-fno-fast-math"]
FAST = ["-ffp-contract=fast"]

# file -> extra flags
SOURCES = {
    "core.hip": [],
    "label_kernels.hip": EXACT,
    "par_kernels.hip": EXACT,
    "permuto_kernels.hip": EXACT,
    "eval_kernels.hip": EXACT,
    "gmm_kernels.hip": EXACT,
    "aug_kernels.hip": EXACT,
    "vit_kernels.hip": FAST,
    "split_kernels.hip": ["-ffp-contract=off"],
    "gemm_kernels.hip": FAST,
    "attn_kernels.hip": FAST + ["-fno-honor-nans"],    # drops the canonicalising v_max the compiler puts in front of fmaxf
    "loss_kernels.hip": FAST,
    # the same two files with fp16 operands (entry points *_f16, cosa_amd/csrc/op16.hpp)
    "gemm_kernels.hip@f16": FAST + ["-DCOSA_OP_F16=1"],
    "attn_kernels.hip@f16": FAST + ["-fno-honor-nans", "-DCOSA_OP_F16=1"],
    # the lattice file once more with a 2-D lattice: the position-only Gaussian kernel of the dense-CRF post-processing
    "permuto_kernels.hip@d2": EXACT + ["-DCOSA_PD=2"],
    "optim_kernels.hip": ["-ffp-contract=off"],
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build_all(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "cosa_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objs, relink, procs = [], force, []
    for key, extra in SOURCES.items():
        src, _, tag = key.partition("@")
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJDIR, src.replace(".hip", ("_" + tag if tag else "") + ".o"))
        objs.append(op)
        extra = extra + os.environ.get("COSA_EXTRA_FLAGS_" + src.split(".")[0].upper(), "").split()     # experiments: per-file extra flags
        if force or _newer(sp, op) or os.path.getmtime(op) < hdr_time:
            cmd = [hipcc, "-c", sp, "-o", op] + COMMON + extra
            if verbose:
                print(" ".join(cmd))
            procs.append((key, subprocess.Popen(cmd)))
            relink = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if relink or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
