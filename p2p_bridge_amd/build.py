"""Builds libp2pb_hip.so (the C-ABI library of include/p2pb_hip.h) for gfx950 with hipcc, in-tree.

    python -m p2p_bridge_amd.build [--force]

hipcc cross-compiles without a GPU. One object per .hip (conv3d.hip: two, built in parallel -- its header) so edits rebuild
in seconds to minutes; a build from scratch takes about 2.5 minutes on 8 cores.
-ffp-contract=off: every fused multiply-add is spelled __fmaf_rn in the sources (arithmetic contract,
DESIGN.md); the compiler must not invent others. -munsafe-fp-atomics: fp32 atomicAdd -> one
global_atomic_add_f32 instead of a CAS loop.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libp2pb_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


# conv3d.hip is three translation units (its header): the second holds the bf16x6 instantiations of the split kernels, the third
# the bf16x3 ones of the training data gradient
EXTRA_UNITS = {"conv3d.hip": [("_bf16x6", ["-DCONV_TU=6"]), ("_bf16x3", ["-DCONV_TU=3"])]}


def _newer(a, bs):
    return (not os.path.exists(a)) or any(os.path.getmtime(a) < os.path.getmtime(b) for b in bs)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "p2pb_hip.h")]
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])
        for tag, defs in EXTRA_UNITS.get(os.path.basename(s), ()):  # the same source compiled again with other macros
            o = os.path.join(OBJ, os.path.basename(s)[:-4] + tag + ".o")
            objs.append(o)
            if force or _newer(o, [s] + hdrs):
                jobs.append([HIPCC] + FLAGS + defs + ["-c", s, "-o", o])
    jobs.sort(key=lambda j: 0 if "conv3d" in j[-3] else 1)  # the two long compiles first

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
