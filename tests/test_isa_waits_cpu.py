"""ISA-level guard for the hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA rings (gemm8.hip, xslin.hip, tfused.hip, attention.hip's
flash3_kernel).

Those kernels keep LDS-DMA in flight across raw `s_barrier`s and retire it with COUNTED waits: gfx950's vmcnt counts every
vector-memory instruction of a wave (loads, stores, LDS-DMA) in issue order, so a wait is right only while the number of such
instructions issued per ring step is exactly what the count was derived from -- and only while hipcc emits ONE instruction per
builtin and no ordinary vector load of its own beside them (for those it waits vmcnt(0) and drains the ring; a load the compiler
hoists or synthesises can also shift the count).  A miscount is a timing-dependent wrong answer, not a crash.

The test disassembles the shipped objects (llvm-objdump on the gfx950 code object inside each host object) and checks, per
instantiation: the LDS-DMA / load / store instructions of the ring loop and of the prologue against the numbers the waits were
derived for, and the wait immediates against the kernel's own formulas.  Adding or removing one DMA issue (or a compiler change
that does) fails here -- re-derive the waits, then update the table.  CPU only: hipcc cross-compiles, nothing runs."""
import collections
import os
import re

import pytest

import isa_util as I

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "mvoc_amd", "csrc")


@pytest.fixture(scope="module")
def objs():
    from mvoc_amd import build
    build.build(verbose=False)
    return {f: I.disassemble(os.path.join(CSRC, f + ".o")) for f in ("gemm8", "xslin", "tfused", "attention")}


def _targs(name):
    return tuple(int(x) for x in re.findall(r"L[ib](\d+)E", name))


def _k_loop(ins, per_iter_mfma):
    """the smallest loop holding exactly the MFMAs of one K-loop iteration"""
    cands = [(a, b) for a, b in I.loops(ins) if sum(1 for i in ins[a:b + 1] if i.is_mfma) == per_iter_mfma]
    assert cands, "K loop not found"
    return min(cands, key=lambda ab: ab[1] - ab[0])


def test_gemm8_counted_waits_match_the_dma_issued(objs):
    """gemm8.hip: per K tile a wave issues 2 + 2 activation pieces (Yh0, Yh1) and PX + PX weight pieces (PX = its share of the
    XPC = 16 / 20 one-KB pieces of an X half-tile over 8 waves); `wait_tile` leaves {Yh0, Xh0, Yh1} of the tile after next in
    flight: 2 + PX + 2."""
    ks = {n: v for n, v in objs["gemm8"].items() if "gemm8_kernel" in n}
    assert len(ks) == 9  # 4 epilogue forms of the 256-wide tile, 3 of the 320-wide, the folded-upsample form (plain epilogue only),
    #                      the chunk-major K form of the 320-wide tile (round 6: KO, plain epilogue only)
    for name, ins in ks.items():
        targs = _targs(name)
        xt, ko = targs[0], len(targs) > 6 and targs[6] == 1
        xpc = xt * 4                                   # 1 KB pieces per X half-tile
        px = sorted({xpc // 8, -(-xpc // 8)})          # per wave
        a, b = _k_loop(ins, 2 * 4 * xt * 4)            # two K tiles x four phases x (XT x 2 x 2) MFMAs
        loop, pro = ins[a:b + 1], ins[:a]
        # the K loop issues LDS-DMA and nothing else on the vector-memory queue
        assert [i.op for i in loop if i.is_vmem and not i.is_lds_dma] == [], name
        assert [i.op for i in pro if i.is_vmem and not i.is_lds_dma] == [], (name, "an ordinary load in the prologue drains the ring")
        n_dma = sum(1 for i in loop if i.is_lds_dma)
        # static instructions of two K tiles: 2 x (4 Y + 2 x PXmax X); the 320-wide tile's third piece is one predicated instruction
        assert n_dma == {4: 16, 5: 18}[xt], (name, n_dma, ko)
        waits = collections.Counter(i.vmcnt() for i in loop if i.vmcnt() is not None)
        want = {0: 2}                                  # (t + 2 == nk: the tail drains) once per K tile body
        for p_ in px:
            want[2 + p_ + 2] = 2
        assert dict(waits) == want, (name, dict(waits), want)
        # prologue: tile 0 complete + {Yh0, Xh0, Yh1} of tile 1 (+ the 256-wide tile's epilogue-table piece), same two waits
        assert sum(1 for i in pro if i.is_lds_dma) == {4: 15, 5: 16}[xt], name
        assert sorted(set(i.vmcnt() for i in pro if i.vmcnt() is not None)) == [0] + [2 + p_ + 2 for p_ in px], name


# (LDS-DMA instructions, stores) in the stage loop of every instantiation, as compiled by the image's hipcc (ROCm 7.2); the
# loops are partly rolled, so these are static counts -- what matters is that they do not move unnoticed
XSLIN_PINS = {  # (NK, RG, ACT) -> (dma, stores)
    (20, 1, 0): (3, 1), (20, 1, 1): (11, 2), (20, 1, 2): (3, 1), (20, 1, 3): (3, 1),
    (20, 2, 0): (1, 4), (20, 2, 1): (2, 3), (20, 2, 2): (1, 4), (20, 2, 3): (1, 4),
    (8, 1, 0): (3, 1), (8, 1, 1): (6, 0), (8, 1, 2): (3, 1), (8, 1, 3): (3, 1),
    (8, 2, 0): (7, 4), (8, 2, 1): (2, 3), (8, 2, 2): (1, 4), (8, 2, 3): (1, 4),
    (4, 1, 0): (4, 1), (4, 1, 1): (8, 0), (4, 1, 2): (4, 1), (4, 1, 3): (4, 1),
    (4, 2, 0): (2, 2), (4, 2, 1): (4, 3), (4, 2, 2): (2, 4), (4, 2, 3): (2, 4),
}
TFUSED_PINS = {(20, 4): (6, 2), (20, 8): (6, 2), (8, 4): (12, 2), (8, 8): (6, 2), (4, 4): (6, 2), (4, 8): (6, 2)}  # (NK, NW)


def test_xslin_ring_has_no_ordinary_loads_and_its_steady_waits(objs):
    """xslin.hip: weights, constants and residual tiles all arrive by LDS-DMA; steady-state waits are PW + {2, 4, 8} RG
    (GEGLU / plain / residual) for the two PW a wave can have (pieces NP = NK + 1 over 4 waves)"""
    ks = {_targs(n): v for n, v in objs["xslin"].items() if "xslin_kernel" in n}
    assert set(ks) == set(XSLIN_PINS)
    for key, ins in ks.items():
        nk, rg, act = key
        loop = I.main_loop(ins)
        assert [i.op for i in loop if i.is_vmem and not i.is_lds_dma and not i.is_store] == [], key
        got = (sum(1 for i in loop if i.is_lds_dma), sum(1 for i in loop if i.is_store))
        assert got == XSLIN_PINS[key], (key, got)
        np_ = nk + 1
        pws = {np_ // 4, -(-np_ // 4)}
        waits = {i.vmcnt() for i in ins if i.vmcnt() is not None}  # (whole kernel: hipcc peels / rotates the stage loop per variant)
        mult = (2,) if act == 1 else (4, 8)            # GEGLU stores after the gate tile only and carries no residual
        for pw in pws:
            for k in mult:
                assert pw + k * rg in waits or pw + k * rg > 22, (key, pw, k, sorted(waits))


def test_tfused_ring_has_no_ordinary_loads(objs):
    """tfused.hip: the weight stream is LDS-DMA only; per-head output stores are the only other vector-memory instructions"""
    ks = {_targs(n): v for n, v in objs["tfused"].items() if "tfused_kernel" in n}
    assert set(ks) == set(TFUSED_PINS)
    for key, ins in ks.items():
        loop = I.main_loop(ins)
        assert [i.op for i in loop if i.is_vmem and not i.is_lds_dma and not i.is_store] == [], key
        got = (sum(1 for i in loop if i.is_lds_dma), sum(1 for i in loop if i.is_store))
        assert got == TFUSED_PINS[key], (key, got)
        nk, nw = key
        pw = -(-nk // nw)
        waits = {i.vmcnt() for i in ins if i.vmcnt() is not None}
        if nw == 4 and nk % nw == 0:                   # the exact steady-state forms: pieces + {2, 4} output stores behind them
            assert {pw + 2, pw + 4} <= waits, (key, sorted(waits))


def test_flash3_ring_waits_match_the_pieces_issued(objs):
    """attention.hip flash3_kernel<NV>: per key tile a wave issues 2 K + 2 NV V pieces (asm statements: hipcc does not see them,
    so EVERY vmcnt wait of the loop is the kernel's own) into a ring of three stages; the wait at the top of an iteration leaves
    (stages - 3) tiles in flight.  The loop holds two tiles (the score accumulators swap roles statically)."""
    ks = {_targs(n)[0]: v for n, v in objs["attention"].items() if "flash3_kernel" in n}
    assert set(ks) == {1, 2}
    for nv, ins in ks.items():
        pcs = 2 + 2 * nv
        a, b = I.main_loop_bounds(ins)
        loop = ins[a:b + 1]
        assert sum(1 for i in loop if i.is_mfma) == 2 * (8 + 8 * nv), nv
        assert sum(1 for i in loop if i.is_lds_dma) == 2 * pcs, nv
        assert [i.op for i in loop if i.is_vmem and not i.is_lds_dma] == [], (nv, "an ordinary vector-memory instruction inside the ring loop")
        waits = [i.vmcnt() for i in loop if i.vmcnt() is not None]
        assert waits == [0, 0], (nv, waits)  # three stages: tile t + 1 is the newest in flight at the top of iteration t
        # prologue: two tiles out, the first one complete -> one tile's pieces may remain; every LDS-DMA has M0 written just ahead
        pro = ins[:a]
        assert sum(1 for i in pro if i.is_lds_dma) == 2 * pcs, nv
        assert pcs in [i.vmcnt() for i in pro if i.vmcnt() is not None], nv
        for j, i in enumerate(ins):
            if i.is_lds_dma:
                assert ins[j - 2].op == "s_mov_b32" and ins[j - 2].args.startswith("m0,") and ins[j - 1].op == "s_nop", (nv, hex(i.addr))
        # the epilogue drains the pieces issued past the last tile before the block's LDS is released
        assert 0 in [i.vmcnt() for i in ins[b + 1:] if i.vmcnt() is not None], nv


def test_no_waterfall_loop_wraps_a_dma(objs):
    """hipcc wraps a buffer / LDS-DMA instruction whose descriptor it cannot prove wave-uniform in a readfirstlane loop (one
    instruction per DISTINCT descriptor value: the count becomes data dependent).  None of the three kernels may contain one."""
    for f, ks in objs.items():
        for name, ins in ks.items():
            if "kernel" not in name:
                continue
            for a, b in I.loops(ins):
                seg = ins[a:b + 1]
                if b - a < 40 and any(i.is_lds_dma for i in seg) and any(i.op.startswith("v_readfirstlane") for i in seg) and \
                        any(i.op.startswith("s_and_saveexec") for i in seg):
                    raise AssertionError(f"{f}: waterfall loop around an LDS-DMA in {name} at {ins[a].addr:#x}")
