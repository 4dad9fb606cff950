"""Build libavi_talking_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

The library is built in-tree (``avi-talking_amd/libavi_talking_hip.so``) so it travels with the
snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libavi_talking_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# No packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Measured on MI355X: a kernel whose
# waves execute them returns wrong values (one half of a register pair, across the lanes of a wave) while waves of a
# matrix-core kernel launched from ANOTHER stream share its SIMDs; alone, serialised, or built without these
# instructions the same kernel is bit-exact (scripts/diag_concurrency.py: 100 % of runs wrong -> 0 %).  The sampling
# pipeline overlaps a second stream with the audio branch, so every translation unit is built without them; the cost
# is below the run-to-run noise of the step (AVI_PACKED_FP32=1 restores them for the diagnostic).
if os.environ.get("AVI_PACKED_FP32", "0") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS += os.environ.get("AVI_DEFINES", "").split()      # e.g. -DAVI_PP_STAMPS for scripts/pp_stamps.py


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "avi_talking.h"))
    # objects compiled with other flags (e.g. a diagnostic AVI_PACKED_FP32=1 / AVI_DEFINES build) are never reused
    stamp = os.path.join(objdir, "flags.txt")
    flags = " ".join([HIPCC] + FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    with open(stamp, "w") as fh:
        fh.write(flags)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
