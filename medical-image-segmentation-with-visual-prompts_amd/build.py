"""Compile the HIP sources in csrc/ for gfx950 into libmivp_hip.so (in-tree).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmivp_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]


# swin_bwd_fused.hip: the SLP vectoriser packs the four dS = P * dP multiplies of a tile into two v_pk_mul_f32 and then
# spends four v_mov per tile re-pairing their operands (the exponentials land in another order): 6 VALU instructions
# instead of 4 in a VALU-issue-bound loop
PER_FILE_FLAGS = {"swin_bwd_fused.hip": ["-fno-slp-vectorize"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(os.path.dirname(HERE), "include", "mivp.h")]
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = ["hipcc"] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        return s

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print("compiled", os.path.basename(s), flush=True)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or _stale(LIB, objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
