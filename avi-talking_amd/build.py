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
# The mechanism inside the hardware is not established: tests/test_cabi_and_host.py disassembles the built
# library and asserts that no such instruction is present, and lib.load() refuses a diagnostic build.
if os.environ.get("AVI_PACKED_FP32", "0") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
else:
    FLAGS += ["-DAVI_BUILD_PACKED_FP32=1"]
# the target feature is meaningless to the x86 host pass of hipcc, which says so once per translation unit
HOST_PASS_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
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
            procs.append((src, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for src, p in procs:
        err = p.communicate()[1]
        err = "".join(l for l in err.splitlines(True) if HOST_PASS_NOISE not in l)
        if err.strip():
            sys.stderr.write(err)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    with open(stamp, "w") as fh:
        fh.write(flags)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def code_objects(lib=LIB):
    """The gfx950 code objects (ELF images) embedded in the shared library: walks the clang offload bundles of its
    .hip_fatbin section (magic, entry count, then (offset, size, triple) records)."""
    import struct
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(magic))[0]
        off = i + len(magic) + 8
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple and sz:
                out.append(data[i + o:i + o + sz])
        pos = i + 1


def instruction_census(lib=LIB, patterns=("v_mfma", r"v_pk_(fma|mul|add)_f32")):
    """Counts of the instructions matching each regular expression over every gfx950 code object of the library
    (llvm-objdump -d): {pattern: count}, plus the number of code objects under "code_objects"."""
    import re
    import tempfile
    counts = {p: 0 for p in patterns}
    objs = code_objects(lib)
    for blob in objs:
        with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
            fh.write(blob)
            fh.flush()
            asm = subprocess.run([OBJDUMP, "-d", fh.name], capture_output=True, text=True, check=True).stdout
        for p in patterns:
            counts[p] += len(re.findall(r"^\s*" + p, asm, flags=re.M))
    counts["code_objects"] = len(objs)
    return counts


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--census" in sys.argv:
        print(instruction_census())
