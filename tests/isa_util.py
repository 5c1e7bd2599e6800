"""Disassembly helpers for the ISA-level tests (test infrastructure): extract the gfx950 code object from a host object that
hipcc produced, disassemble it with llvm-objdump and hand back, per kernel, the instruction list and its loops.

Used by tests/test_isa_waits_cpu.py to pin the hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA rings (gemm8.hip, xslin.hip,
tfused.hip) to the number of vector-memory instructions the shipped code object really issues per ring step."""
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_TARGET = re.compile(r"<[^>+]+\+0x([0-9a-f]+)>\s*$")


class Ins:
    __slots__ = ("addr", "op", "args", "target")

    def __init__(self, addr, op, args, target):
        self.addr, self.op, self.args, self.target = addr, op, args, target

    @property
    def is_lds_dma(self):
        return (self.op.startswith("buffer_load") or self.op.startswith("global_load_lds")) and (" lds" in " " + self.args or self.op.startswith("global_load_lds"))

    @property
    def is_vmem(self):
        return self.op.startswith(("buffer_", "global_", "flat_", "scratch_")) and not self.op.startswith(("buffer_wbl2", "buffer_inv"))

    @property
    def is_store(self):
        return self.is_vmem and "_store" in self.op

    @property
    def is_mfma(self):
        return self.op.startswith("v_mfma")

    def vmcnt(self):
        """the N of an `s_waitcnt ... vmcnt(N)` instruction, else None"""
        if self.op != "s_waitcnt":
            return None
        m = re.search(r"vmcnt\((\d+)\)", self.args)
        return int(m.group(1)) if m else None

    def __repr__(self):
        return f"{self.addr:#x}: {self.op} {self.args}"


def disassemble(obj_path):
    """host object (with an embedded gfx950 offload bundle) -> {kernel symbol: [Ins]}"""
    tmp = tempfile.mkdtemp(prefix="mvoc_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        assert co, f"no gfx950 bundle in {obj_path}"
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co[0])], check=True, capture_output=True,
                              text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m:
            base = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = _INS.match(line)
        if m and cur is not None:
            op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
            tgt = None
            if op.startswith(("s_cbranch", "s_branch")):
                t = _TARGET.search(line)
                if t:
                    tgt = base + int(t.group(1), 16)
            cur.append(Ins(addr, op, args, tgt))
    return kernels


def loops(ins):
    """[(first index, last index)] of every backward branch's span, outermost first by size"""
    idx = {i.addr: n for n, i in enumerate(ins)}
    out = []
    for n, i in enumerate(ins):
        if i.target is not None and i.target <= i.addr and i.target in idx:
            out.append((idx[i.target], n))
    return sorted(out, key=lambda ab: ab[0] - ab[1])


def main_loop(ins):
    """the loop with the most MFMAs among the loops that contain no other MFMA-carrying loop (the K loop / stage loop)"""
    a, b = main_loop_bounds(ins)
    return ins[a:b + 1]


def main_loop_bounds(ins):
    """(first index, last index) of main_loop"""
    ls = loops(ins)
    best, best_n = None, -1
    for a, b in ls:
        n = sum(1 for i in ins[a:b + 1] if i.is_mfma)
        inner = any((a2, b2) != (a, b) and a <= a2 and b2 <= b and any(i.is_mfma for i in ins[a2:b2 + 1]) for a2, b2 in ls)
        if n > best_n and not inner:
            best, best_n = (a, b), n
    assert best is not None, "no MFMA loop"
    return best
