"""In-tree build of libssbev_hip.so (hipcc, gfx950 only).  `python -m stereoscene_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libssbev_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


# Kernels whose results must be BIT-identical to the CPU oracle are built without FMA contraction
# (hipcc's __fmul_rn/__fadd_rn are plain * and + and would otherwise fuse).
PER_FILE_FLAGS = {"voxel_pool.hip": ["-ffp-contract=off"], "lidar_depth.hip": ["-ffp-contract=off"],
                  "bri_shell.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=False):
    """tuning=True (`--tuning`): -DSSBEV_TUNING, the ~50 kernel tuning hooks of csrc/ (ssbev_tune in common.h) answer the environment;
    the product build compiles them out.  Switching between the two needs --force."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "ssbev.h"))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc, *FLAGS, *(["-DSSBEV_TUNING"] if tuning else []), *PER_FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv or "--tuning" in sys.argv, tuning="--tuning" in sys.argv)
    print(LIB)
