"""Build libmvoc_hip.so (gfx950) in-tree with hipcc.  No CMake, no JIT cache: the .so sits next to the sources so
it travels to the GPU box with the repo snapshot."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmvoc_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "gemm8.hip", "attention.hip", "tfused.hip", "xslin.hip", "norm.hip", "pnp.hip", "stem.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LAB = os.environ.get("MVOC_BUILD_LAB") == "1"  # diagnostic library (in-kernel stamps / ablations): libmvoc_hip_lab.so
if LAB:
    LIB = os.path.join(HERE, "libmvoc_hip_lab.so")
# A/B builds of one experiment: MVOC_BUILD_VARIANT=<name> MVOC_BUILD_DEFS="-DFOO=1 ..." -> libmvoc_hip_<name>.so beside the product library
# (objects <src>.<name>.o); loaded with MVOC_HIP_LIB=<path> (mvoc_amd/_ffi.py).  Never the product build.
VARIANT = os.environ.get("MVOC_BUILD_VARIANT", "")
VARIANT_DEFS = os.environ.get("MVOC_BUILD_DEFS", "").split()
if VARIANT:
    LIB = os.path.join(HERE, f"libmvoc_hip_{VARIANT}.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage"]  # the remarks are kept per object (<obj>.res.txt): tests check them
# attention: keep the MFMA accumulators in VGPRs (gfx950 has one unified register file).  In AGPR form hipcc time-shares
# 32 AGPRs between the S^T and O^T accumulators and emits ~220 v_accvgpr_read/write per K/V tile; VGPR form has none
# and needs 162 instead of 196 registers (3 waves per SIMD instead of 2).
# gemm: neutral for the 32-row-per-wave tiles (same speed, no accvgpr traffic), required by the 64-row-per-wave tiles
# (160 / 128 accumulator registers + operands fit 256 unified registers only in this form -> 2 waves per SIMD).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # gemm8: loop headers on 64-byte boundaries -- the K loop's speed depends on where its first instruction falls (the
               # guide's code-placement note); pinned so that edits elsewhere in the kernel do not move it
               "gemm8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-falign-loops=64"], "tfused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "xslin.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _digest():
    h = hashlib.sha256()
    # sources only (.hip / .h): objects and their stamps live in the same directory and must not feed the digest
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith((".hip", ".h"))) + ["../../include/mvoc_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update((" ".join(FLAGS + VARIANT_DEFS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    procs = []
    shared = b"".join(open(os.path.join(CSRC, f), "rb").read() for f in sorted(os.listdir(CSRC)) if f.endswith(".h"))
    shared += open(os.path.join(HERE, "..", "include", "mvoc_hip.h"), "rb").read()
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", f".{VARIANT}.o" if VARIANT else (".lab.o" if LAB else ".o")))
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, *(["-DMVOC_PP_LAB"] if LAB else []), *VARIANT_DEFS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        # per-object digest (source + every header + the command line): unchanged translation units are not recompiled
        od = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + shared + " ".join(cmd).encode()).hexdigest()
        ostamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == od:
            continue
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), ostamp, od))
    failed = False
    for src, p, ostamp, od in procs:
        out, _ = p.communicate()
        remarks = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l)
        out = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l)
        with open(ostamp.replace(".stamp", ".res.txt"), "w") as fh:  # registers / scratch / LDS / occupancy of every kernel
            fh.write(remarks + "\n")
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}:\n{out}", file=sys.stderr)
        else:
            with open(ostamp, "w") as fh:
                fh.write(od)
    if failed:
        raise RuntimeError("libmvoc_hip.so: compilation failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
