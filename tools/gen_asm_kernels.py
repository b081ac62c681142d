#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 kernel bodies (ark-mpc_amd/csrc/asm_kernels.inc).

Why: hipcc's code for the 256-bit Montgomery core spends ~25 % of its VALU issue slots on v_mov (64-bit
register-pair shuffling around v_mad_u64_u32), cannot fuse sums of products before one reduction, and delays
half of each 64-byte record's loads until the line has left L2.  The bodies emitted here keep every operand
in fixed VGPRs, issue all loads up front, use one carry chain per multiplier row with the NEXT row's
multiplies interleaved into it, and do lazy reduction:  K3 = 6 products but only 3 Montgomery reductions.

hipcc neither schedules nor pads inside an asm statement, so the hazard bookkeeping lives here:
  H1  VALU writes SGPR/VCC -> VALU reads it (carry-in, v_cndmask mask): >= 2 wait states in between
      (gfx940-class; hipcc pads its own code the same way).  Tracked by `Emitter`, filled with s_nop.
  H2  a global_load's VGPRs are read only after an s_waitcnt vmcnt(N) that covers it (in-order return)
  H3  constant-bus: an instruction reading VCC as carry has only VGPR / inline-constant sources
  H4  nothing is written after the stores; loaded-over registers are dead before the load is issued
VGPR read-after-write between VALU ops is interlocked by hardware.

Every stream is first run through the single-lane emulator below against Python big-int arithmetic
(`--selftest`, also tests/test_asm_generator.py), then parity-tested on the GPU against the oracle.

Semantics: online-phase/src/algebra/scalar/authenticated_scalar.rs:161-171 (combine), :871-878
(de + d[b] + e[a] + [c]); share.rs:74-77 (add_public adds to the share only for PARTY0).
"""
import argparse
import os
import random
import sys

FIELDS = [
    ("BN254_FR", 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001),
    ("BLS12_381_FR", 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001),
    ("CURVE25519_FR", 2**252 + 27742317777372353535851937790883648493),
    ("BN254_FQ", 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47),
    ("CURVE25519_FQ", 2**255 - 19),
]
COORD_ONLY = {"CURVE25519_FQ"}   # 2p^2 overflows the 9-limb lazy accumulator; only the single-multiplication block is emitted
M32 = 0xFFFFFFFF
R = 1 << 256
JUNK = "s[60:61]"     # carry-out sink of v_mad_u64_u32 (never read)
S_INV = "s62"         # -p^{-1} mod 2^32
CLOBBER_SGPRS = ["s60", "s61", "s62", "s64", "s65"]


class Ins:
    """One instruction: assembler text + emulator op + every register it reads / writes (for the scheduler).
    srd / swr are the SGPR-class registers (carry / mask) subject to hazard H1."""
    __slots__ = ("text", "op", "args", "srd", "swr", "rd", "wr", "idx")

    def __init__(self, text, op, args=(), srd=(), swr=(), rd=(), wr=()):
        self.text, self.op, self.args, self.srd, self.swr = text, op, args, tuple(srd), tuple(swr)
        self.rd = set(rd) | set(srd)
        self.wr = set(wr) | set(swr)
        self.idx = -1


def pr(pair):
    a, b = int(pair[0][1:]), int(pair[1][1:])
    assert b == a + 1 and a % 2 == 0, "64-bit operand needs an even-aligned register pair: %r" % (pair,)
    return "v[%d:%d]" % (a, b)


def quad(regs4):
    a = int(regs4[0][1:])
    assert a % 4 == 0 and [int(r[1:]) for r in regs4] == [a, a + 1, a + 2, a + 3]
    return "v[%d:%d]" % (a, a + 3)


def src(x):
    if isinstance(x, int):
        return str(x) if 0 <= x <= 64 else "0x%08x" % x
    return x


def regs_of(*xs):
    """register names among operands (ints are inline constants / literals)"""
    out = []
    for x in xs:
        if isinstance(x, tuple):
            out += list(x)
        elif isinstance(x, str):
            out.append(x)
    return out


CY2 = "s[64:65]"      # second carry register (VOP3 encodings); VCC is the first


# ---- instruction constructors.  cy = "vcc" (VOP2 encodings) or an SGPR pair (VOP3 encodings) -------------
def i_mov(d, s): return Ins("v_mov_b32_e32 %s, %s" % (d, src(s)), "mov", (d, s), rd=regs_of(s), wr=[d])
def i_mad(d, a, b, c):
    return Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(d), JUNK, a, b, "0" if c == 0 else pr(c)), "mad", (d, a, b, c),
               rd=regs_of(a, b, c if c != 0 else None), wr=list(d))
def i_mul_lo(d, a, b): return Ins("v_mul_lo_u32 %s, %s, %s" % (d, a, b), "mul_lo", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_addco(d, a, b, cy="vcc"):
    t = "v_add_co_u32_e32 %s, vcc, %s, %s" % (d, src(a), b) if cy == "vcc" else "v_add_co_u32_e64 %s, %s, %s, %s" % (d, cy, src(a), b)
    return Ins(t, "addco", (d, a, b, cy), swr=(cy,), rd=regs_of(a, b), wr=[d])
def i_addc(d, a, b, cy="vcc"):
    t = "v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (d, src(a), b) if cy == "vcc" else "v_addc_co_u32_e64 %s, %s, %s, %s, %s" % (d, cy, src(a), b, cy)
    return Ins(t, "addc", (d, a, b, cy), srd=(cy,), swr=(cy,), rd=regs_of(a, b), wr=[d])
def i_subco(d, a, b, cy="vcc"):
    t = "v_sub_co_u32_e32 %s, vcc, %s, %s" % (d, src(a), b) if cy == "vcc" else "v_sub_co_u32_e64 %s, %s, %s, %s" % (d, cy, src(a), b)
    return Ins(t, "subco", (d, a, b, cy), swr=(cy,), rd=regs_of(a, b), wr=[d])
def i_subb(d, a, b, cy="vcc"):
    t = "v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (d, src(a), b) if cy == "vcc" else "v_subb_co_u32_e64 %s, %s, %s, %s, %s" % (d, cy, src(a), b, cy)
    return Ins(t, "subb", (d, a, b, cy), srd=(cy,), swr=(cy,), rd=regs_of(a, b), wr=[d])
def i_cnd(d, f, t, cy="vcc"):   # cy ? t : f
    tx = "v_cndmask_b32_e32 %s, %s, %s, vcc" % (d, f, t) if cy == "vcc" else "v_cndmask_b32_e64 %s, %s, %s, %s" % (d, f, t, cy)
    return Ins(tx, "cnd", (d, f, t, cy), srd=(cy,), rd=regs_of(f, t), wr=[d])
def i_and(d, a, b): return Ins("v_and_b32_e32 %s, %s, %s" % (d, src(a), b), "and", (d, a, b), rd=regs_of(a, b), wr=[d])
# Streams that are read / written exactly once per batch get the non-temporal hint in the split-column layout, so they
# do not displace the lines K3 re-reads after K1 (own/peer d||e, a.s, b.s) from the 256 MiB Infinity Cache.
# Measured on MI355X (2^20 gates, split layout): K3 80 -> 69 us.  In the AoS layout share and MAC halves share one
# 64-byte line and the hint hurts, so the AoS variant carries none.
NT_SPLIT = {"c_s", "c_m", "out_s", "out_m", "a_m", "b_m"}
NT_AOS = set(os.environ.get("ARKMPC_GEN_NT_AOS", "").split(",")) - {""}
NT_BASES = set()
# LDS-staged AoS variant: the workgroup has copied its 256 a / b / c records (64 B each) into LDS with coalesced non-temporal
# loads -- every 64-byte line consumed by ONE instruction, which is what makes the hint usable for AoS data -- and the body
# reads its own record from there; the result record goes back through the a-region.  Record layout: share at +0, MAC at +32.
LDS_MODE = False
# split-layout variant with the RESULT staged through LDS (global loads as usual): a thread leaves its share at lds_off and its MAC at
# lds_off + 8192 (lds_off = 32 * thread), and the workgroup streams both 8 KiB columns out with lane-contiguous non-temporal stores --
# whole lines per instruction.  Non-temporal 16-byte stores 32 B apart (what the body issues otherwise) leave the L2 as partial lines:
# +10 % write traffic on K3, +23..34 % on pure store kernels (probes/write_calib.hip).
LDS_OUT_MODE = False
LDS_STREAM = {"a": 0, "b": 16384, "c": 32768, "out": 0}
def lds_offset(base, half):
    stream, part = base.split("_")
    return LDS_STREAM[stream] + (32 if part == "m" else 0) + 16 * half


def i_load(regs, off, base, half):
    if LDS_MODE and base.split("_")[0] in ("a", "b", "c"):
        return Ins("ds_read_b128 %s, %%[lds_off] offset:%d" % (quad(regs), lds_offset(base, half)), "load", (regs, base, half),
                   rd=["MEMORDER"], wr=list(regs) + ["MEMORDER"])
    return Ins("global_load_dwordx4 %s, %%[%s], %%[%s]%s%s" % (quad(regs), off, base, " offset:16" if half else "", " nt" if base in NT_BASES else ""), "load", (regs, base, half),
               rd=["MEMORDER"], wr=list(regs) + ["MEMORDER"])
def i_store(regs, off, base, half):
    if LDS_OUT_MODE:
        return Ins("ds_write_b128 %%[lds_off], %s offset:%d" % (quad(regs), (8192 if base.endswith("_m") else 0) + 16 * half), "store", (regs, base, half),
                   rd=list(regs) + ["MEMORDER"], wr=["MEMORDER"])
    if LDS_MODE:
        return Ins("ds_write_b128 %%[lds_off], %s offset:%d" % (quad(regs), lds_offset(base, half)), "store", (regs, base, half),
                   rd=list(regs) + ["MEMORDER"], wr=["MEMORDER"])
    return Ins("global_store_dwordx4 %%[%s], %s, %%[%s]%s%s" % (off, quad(regs), base, " offset:16" if half else "", " nt" if base in NT_BASES else ""), "store", (regs, base, half),
               rd=list(regs) + ["MEMORDER"], wr=["MEMORDER"])
def i_wait(n): return Ins("s_waitcnt vmcnt(%d)" % n, "wait", (n,))
def i_wait_lds(n): return Ins("s_waitcnt lgkmcnt(%d)" % n, "wait", (n,))


class Emitter:
    """Collects the final instruction order and pads H1."""

    def __init__(self):
        self.lines, self.order, self.slot, self.lastw, self.nops = [], [], 0, {}, 0

    def emit(self, ins):
        need = 0
        for r in ins.srd:
            if r in self.lastw:
                need = max(need, self.lastw[r] + 3 - self.slot)
        if need > 0:
            self.lines.append("s_nop %d" % (need - 1))
            self.slot += need
            self.nops += need
        self.lines.append(ins.text)
        self.order.append(ins)
        for r in ins.swr:
            self.lastw[r] = self.slot
        self.slot += 1

    def emit_all(self, seq):
        for i in seq:
            self.emit(i)

    def schedule(self, seq, window=64):
        """Dependency-aware list scheduling of a straight-line segment: any order that respects register RAW / WAR /
        WAW (VGPRs and carry registers alike) is correct on an in-order-issue machine with interlocked VGPR
        dependencies, so the earliest (in program order) ready instruction that needs no H1 wait states is issued;
        s_nop is emitted only when every ready instruction inside the look-ahead window would violate H1."""
        n = len(seq)
        succ = [[] for _ in range(n)]
        indeg = [0] * n
        last_w, readers = {}, {}
        for i, ins in enumerate(seq):
            deps = set()
            for r in ins.rd:
                if r in last_w:
                    deps.add(last_w[r])
            for r in ins.wr:
                if r in last_w:
                    deps.add(last_w[r])
                for j in readers.get(r, ()):
                    deps.add(j)
            deps.discard(i)
            for j in deps:
                succ[j].append(i)
            indeg[i] = len(deps)
            for r in ins.wr:
                last_w[r] = i
                readers[r] = []
            for r in ins.rd:
                readers.setdefault(r, []).append(i)
        done = [False] * n
        lowest = 0
        remaining = n
        while remaining:
            while done[lowest]:
                lowest += 1
            best, best_need = None, None
            for i in range(lowest, min(n, lowest + window)):
                if done[i] or indeg[i]:
                    continue
                need = 0
                for r in seq[i].srd:
                    if r in self.lastw:
                        need = max(need, self.lastw[r] + 3 - self.slot)
                if need <= 0:
                    best, best_need = i, 0
                    break
                if best is None or need < best_need:
                    best, best_need = i, need
            assert best is not None
            self.emit(seq[best])
            done[best] = True
            remaining -= 1
            for j in succ[best]:
                indeg[j] -= 1

    def raw(self, text, op="raw", args=()):
        self.emit(Ins(text, op, args))


class Regs:
    def __init__(self, first):
        self.next = first

    def one(self):
        r = "v%d" % self.next
        self.next += 1
        return r

    def vec(self, n, align=1):
        while self.next % align:
            self.next += 1
        return [self.one() for _ in range(n)]

    def pair(self):
        return tuple(self.vec(2, 2))


def t_bounds(p, nprod):
    """Worst-case CIOS accumulator values for a sum of nprod products with inputs < p (exact integer bounds):
    returns (max just before a reduction row, steady-state max after an outer step)."""
    t = 0
    before = 0
    for _ in range(64):
        before = t + nprod * (p - 1) * M32
        after = (before + M32 * p) >> 32
        if after == t:
            break
        t = after
    return before, t


def montmul_sum_seq(p, prods, T, Tz, q, m, P, row0=0, final_out=None):
    """Straight-line CIOS evaluation of REDC(sum_k a_k*b_k) (before the final conditional subtraction).
    T[j] is the low half of the pair Tz[j] whose high half holds 0 for the whole kernel.  Rows alternate between
    the two carry registers so that the scheduler can keep two carry chains in flight (next row's chain starts
    while this row's is still running) and hazard H1 is met by useful instructions."""
    nprod = len(prods)
    before, _ = t_bounds(p, nprod)
    assert before < (1 << 288), "accumulator needs a 10th limb for this field / product count"
    seq = []
    first = True
    row = row0
    for r in range(8):
        for (a, b) in prods:
            cy = "vcc" if row % 2 == 0 else CY2
            row += 1
            seq += [i_mad(q[j], a[j], b[r], 0 if first else Tz[j]) for j in range(8)]
            seq += [i_mov(T[0], q[0][0]), i_addco(T[1], q[1][0], q[0][1], cy)]
            seq += [i_addc(T[j], q[j][0], q[j - 1][1], cy) for j in range(2, 8)]
            seq += [i_addc(T[8], 0 if first else T[8], q[7][1], cy)]
            first = False
        cy = "vcc" if row % 2 == 0 else CY2
        row += 1
        seq += [i_mul_lo(m, T[0], S_INV)]
        seq += [i_mad(q[j], m, P[j], Tz[j]) for j in range(8)]
        D = final_out if (final_out is not None and r == 7) else T
        seq += [i_addco(D[0], q[1][0], q[0][1], cy)]
        seq += [i_addc(D[j - 1], q[j][0], q[j - 1][1], cy) for j in range(2, 8)]
        seq += [i_addc(D[7], T[8], q[7][1], cy)]
        if D is T:
            seq += [i_addc(T[8], 0, Tz[8][1], cy)]   # Tz[8][1] holds 0 (src1 must be a VGPR)
    return seq, row


def cond_sub_final(p, nprod, T, P, tmp, out, cy="vcc"):
    """T (9 limbs) < p*(1 + nprod*p/R) + 1 -> canonical `out` by K conditional subtractions."""
    bound = (nprod * (p - 1) * (p - 1) + (R - 1) * p) // R + 1
    K = (bound + p - 1) // p - 1
    assert bound < R, "9-limb final subtraction not implemented"
    seq = []
    cur = T
    for k in range(K):
        last = k == K - 1
        dst = out if last else T
        seq.append(i_subco(tmp[0], cur[0], P[0], cy))
        seq += [i_subb(tmp[j], cur[j], P[j], cy) for j in range(1, 8)]
        seq += [i_cnd(dst[j], tmp[j], cur[j], cy) for j in range(8)]          # borrow -> value < p -> keep cur
        cur = dst
    if K == 0:
        seq += [i_mov(out[j], T[j]) for j in range(8)]
    return seq


def fe_add_seq(P, a, b, out, tmp, cy_add="vcc", cy_sub=CY2):
    """out = a + b mod p for canonical a, b (p < 2^255: no 257th bit). `out` may alias a or b; tmp may not.
    The add chain and the subtract chain use different carry registers so they overlap (skewed by one limb)."""
    seq = [i_addco(tmp[0], a[0], b[0], cy_add)] + [i_addc(tmp[j], a[j], b[j], cy_add) for j in range(1, 8)]
    seq += [i_subco(out[0], tmp[0], P[0], cy_sub)] + [i_subb(out[j], tmp[j], P[j], cy_sub) for j in range(1, 8)]
    seq += [i_cnd(out[j], out[j], tmp[j], cy_sub) for j in range(8)]
    return seq


# ------------------------------------------------------------------------------------------------
# single-lane emulator
# ------------------------------------------------------------------------------------------------
class Emu:
    def __init__(self):
        self.v, self.s, self.c = {}, {}, {}

    def rd(self, x):
        if isinstance(x, int):
            return x & M32
        if x.startswith("v"):
            if x not in self.v:
                raise KeyError("read of uninitialised %s" % x)
            return self.v[x]
        return self.s[x]

    def run(self, order):
        for ins in order:
            op, a = ins.op, ins.args
            if op == "mov":
                self.v[a[0]] = self.rd(a[1])
            elif op == "mad":
                d, x, y, c = a
                cv = 0 if c == 0 else (self.rd(c[0]) | (self.rd(c[1]) << 32))
                r = (self.rd(x) * self.rd(y) + cv) & ((1 << 64) - 1)
                self.v[d[0]], self.v[d[1]] = r & M32, r >> 32
            elif op == "mul_lo":
                self.v[a[0]] = (self.rd(a[1]) * self.rd(a[2])) & M32
            elif op in ("addco", "addc"):
                r = self.rd(a[1]) + self.rd(a[2]) + (self.c[a[3]] if op == "addc" else 0)
                self.v[a[0]], self.c[a[3]] = r & M32, r >> 32
            elif op in ("subco", "subb"):
                r = self.rd(a[1]) - self.rd(a[2]) - (self.c[a[3]] if op == "subb" else 0)
                self.v[a[0]], self.c[a[3]] = r & M32, 1 if r < 0 else 0
            elif op == "cnd":
                self.v[a[0]] = self.rd(a[2]) if self.c[a[3]] else self.rd(a[1])
            elif op == "and":
                self.v[a[0]] = self.rd(a[1]) & self.rd(a[2])
            elif op == "load":
                regs, name, half = a
                val = self.mem[name]
                for i, rg in enumerate(regs):
                    self.v[rg] = (val >> (32 * (4 * half + i))) & M32
            elif op == "smov":
                self.s[a[0]] = a[1]
            elif op in ("raw", "store", "wait"):
                pass
            else:
                raise ValueError(op)

    def setv(self, regs, value):
        for i, r in enumerate(regs):
            self.v[r] = (value >> (32 * i)) & M32

    def getv(self, regs):
        return sum(self.v[r] << (32 * i) for i, r in enumerate(regs))


# ------------------------------------------------------------------------------------------------
# K2+K3 body
# ------------------------------------------------------------------------------------------------
def build_beaver_finish(p, first_vgpr=8, key_names=None, sched=True, nt=False, lds=False, ldsout=False):
    global NT_BASES, LDS_MODE, LDS_OUT_MODE
    NT_BASES = (NT_SPLIT - {"out_s", "out_m"} if nt == 2 else NT_SPLIT) if nt else NT_AOS     # nt = 2: hints on the loads only
    LDS_MODE = lds
    LDS_OUT_MODE = ldsout
    try:
        return _build_beaver_finish(p, first_vgpr, key_names, sched, lds, ldsout)
    finally:
        LDS_MODE = False
        LDS_OUT_MODE = False


def _build_beaver_finish(p, first_vgpr, key_names, sched, lds, ldsout=False):
    """Emit the fused combine + finish body for modulus p.  Returns (Emitter, regmap)."""
    key = key_names or ["%[k" + str(i) + "]" for i in range(8)]
    rg = Regs(first_vgpr)
    P = rg.vec(8)
    dm, dp = rg.vec(8, 4), rg.vec(8, 4)
    em, ep = rg.vec(8, 4), rg.vec(8, 4)
    bs, as_ = rg.vec(8, 4), rg.vec(8, 4)
    bm, am = rg.vec(8, 4), rg.vec(8, 4)
    Tz = [rg.pair() for _ in range(9)]
    T = [t[0] for t in Tz]
    q = [rg.pair() for _ in range(8)]
    qflat = [r for pq in q for r in pq]
    m = rg.one()
    nv = rg.next
    inv = (-pow(p, -1, 1 << 32)) & M32
    E = Emitter()
    run = E.schedule if sched else E.emit_all
    E.raw("s_nop 1")                                                  # H1 guard for SGPR inputs written by VALU
    E.raw("s_mov_b32 %s, 0x%08x" % (S_INV, inv), "smov", (S_INV, inv))
    # ---- all first-wave loads up front (16 x dwordx4), in the order they are needed
    for regs, nm in ((dm, "my_d"), (dp, "peer_d"), (em, "my_e"), (ep, "peer_e")):
        for h in (0, 1):
            E.emit(i_load(regs[4 * h:4 * h + 4], "off_de", nm, h))
    for regs, nm in ((bs, "b_s"), (as_, "a_s"), (bm, "b_m"), (am, "a_m")):
        for h in (0, 1):
            E.emit(i_load(regs[4 * h:4 * h + 4], "off_col", nm, h))
    # constants while the loads fly
    for j in range(8):
        E.emit(i_mov(P[j], (p >> (32 * j)) & M32))
    for t in Tz:
        E.emit(i_mov(t[1], 0))
    # ---- regrouping (exact in the field, so the canonical results are the reference's bit for bit):
    #   share = de [PARTY0] + d b.s + e a.s + c.s = d (e [PARTY0] + b.s) + e a.s + c.s
    #   mac   = key de + d b.m + e a.m + c.m     = d (key e + b.m)     + e a.m + c.m
    # FIVE products and three reductions (key e; two sums of two products) instead of six and three (de; a sum of two; a sum of three --
    # or, for moduli above ~2^254.4, six and FOUR).  The second factor of a product is only the row multiplier: it may be any 256-bit
    # value, so the sums e + b.s and key e + b.m are left unreduced (< 2p) where the reduced sum of products still fits 8 limbs.
    def lazy_ok():
        return (3 * (p - 1) * (p - 1) + (R - 1) * p) // R + 1 < R

    def plus(dst, a_, b_, cy):
        """dst = a_ + b_: unreduced when the modulus leaves room, canonical otherwise; returns (sequence, value weight of dst as a row multiplier)"""
        if lazy_ok():
            return [i_addco(dst[0], a_[0], b_[0], cy)] + [i_addc(dst[j], a_[j], b_[j], cy) for j in range(1, 8)], 2
        return fe_add_seq(P, a_, b_, dst, qflat[8:]), 1
    # ---- segment 1: K2 (d = my_d + peer_d, e = my_e + peer_e) and ke = key * e
    E.emit(i_wait(0 if lds else 8))                                  # LDS variant: the eight d||e loads are the only global loads
    d, e, ke = dm, em, dp
    seg = fe_add_seq(P, dm, dp, dm, qflat[:8]) + fe_add_seq(P, em, ep, em, qflat[8:])
    mm, row = montmul_sum_seq(p, [(key, e)], T, Tz, q, m, P)
    seg += mm + cond_sub_final(p, 1, T, P, qflat, ke)
    run(seg)
    # ---- segment 2: share' = d (e [PARTY0] + b.s) + e a.s (one reduction); then c is loaded over the dead b.s / a.s registers
    E.emit(i_wait_lds(4) if lds else i_wait(4))
    rs = ep
    seg = [i_and(qflat[j], "%[mask]", e[j]) for j in range(8)]
    add, w = plus(bs, bs, qflat[:8], "vcc")
    seg += add
    mm, row = montmul_sum_seq(p, [(d, bs), (e, as_)], T, Tz, q, m, P, row)
    seg += mm + cond_sub_final(p, 1 + w, T, P, qflat, rs)
    cs, cm = bs, as_
    for regs, nm in ((cs, "c_s"), (cm, "c_m")):
        for h in (0, 1):
            seg.append(i_load(regs[4 * h:4 * h + 4], "off_col", nm, h))
    run(seg)
    # ---- segment 3: mac' = d (key e + b.m) + e a.m (one reduction); the c loads' latency hides under these rows
    E.emit(i_wait_lds(4) if lds else i_wait(4))
    rm = bm
    seg, w = plus(bm, bm, ke, CY2)
    mm, row = montmul_sum_seq(p, [(d, bm), (e, am)], T, Tz, q, m, P, row)
    run(seg + mm + cond_sub_final(p, 1 + w, T, P, qflat, rm))
    # ---- segment 4: share = share' + c.s ; mac = mac' + c.m ; stores
    E.emit(i_wait_lds(0) if lds else i_wait(0))
    seg = fe_add_seq(P, rs, cs, rs, qflat[:8]) + fe_add_seq(P, rm, cm, rm, qflat[8:])
    for regs, nm in ((rs, "out_s"), (rm, "out_m")):
        for h in (0, 1):
            seg.append(i_store(regs[4 * h:4 * h + 4], "off_out", nm, h))
    run(seg)
    if lds or ldsout:
        E.emit(i_wait_lds(0))                                         # the caller's barrier must see the result record in LDS
    regmap = dict(P=P, dm=dm, dp=dp, em=em, ep=ep, bs=bs, as_=as_, bm=bm, am=am, rs=rs, rm=rm, first=first_vgpr, nv=nv)
    return E, regmap


# ------------------------------------------------------------------------------------------------
# single Montgomery multiplication as an inline-asm block for C++ callers (EC point formulas)
# ------------------------------------------------------------------------------------------------
MM_FIRST_VGPR = 32            # fixed temporaries v32..v66 (clobbered); operands are compiler-allocated ("v" constraints)
MM_SGPR_P = ["s%d" % (21 + i) for i in range(8)]
MM_CLOBBER_SGPRS = ["s16", "s17", "s18", "s19", "s20"] + MM_SGPR_P


def build_montmul(p):
    """out = a*b*R^-1 mod p as a value in [0, 2p) (all four moduli: 2p < 2^256); the caller canonicalises.
    Operands %[a0..a7], %[b0..b7] (inputs) and %[o0..o7] (early-clobber outputs) are chosen by the compiler."""
    global JUNK, S_INV, CY2
    saved = (JUNK, S_INV, CY2)
    JUNK, S_INV, CY2 = "s[16:17]", "s20", "s[18:19]"
    try:
        # temporaries live ONLY in caller-saved VGPR blocks of the AMDGPU calling convention (v32-39, v48-55, v64-71, v80-87,
        # v96-103; v40-47, v56-63, v72-79, ... are callee-saved): a __noinline__ device function that clobbered callee-saved
        # registers would have to spill and restore them around every call
        blocks = [32, 48, 64, 80, 96]
        pairs = [("v%d" % (b + 2 * k), "v%d" % (b + 2 * k + 1)) for b in blocks for k in range(4)]
        Tz, q = pairs[:9], pairs[9:17]
        T = [t[0] for t in Tz]
        m = pairs[17][0]
        used = sorted({int(r[1:]) for pr_ in pairs[:17] for r in pr_} | {int(m[1:])})
        nv = None
        a = ["%%[a%d]" % i for i in range(8)]
        b = ["%%[b%d]" % i for i in range(8)]
        o = ["%%[o%d]" % i for i in range(8)]
        E = Emitter()
        E.raw("s_nop 1")
        E.raw("s_mov_b32 %s, 0x%08x" % (S_INV, (-pow(p, -1, 1 << 32)) & M32), "smov", (S_INV, (-pow(p, -1, 1 << 32)) & M32))
        for j in range(8):
            E.raw("s_mov_b32 %s, 0x%08x" % (MM_SGPR_P[j], (p >> (32 * j)) & M32), "smov", (MM_SGPR_P[j], (p >> (32 * j)) & M32))
        seq = [i_mov(t[1], 0) for t in Tz]
        mm, _ = montmul_sum_seq(p, [(a, b)], T, Tz, q, m, MM_SGPR_P, 0, final_out=o)
        E.schedule(seq + mm)
        bound = ((p - 1) * (p - 1) + (R - 1) * p) // R + 1
        assert bound < 2 * p and 2 * p < R
        return E, dict(first=MM_FIRST_VGPR, nv=nv, a=a, b=b, o=o, used=used)
    finally:
        JUNK, S_INV, CY2 = saved


def selftest_montmul(p, trials=200, seed=3):
    rng = random.Random(seed)
    E, mp = build_montmul(p)
    Rinv = pow(R, -1, p)
    edge = [0, 1, p - 1, p - 2, R % p, (p + 1) // 2, (1 << 255) % p]
    for t in range(trials):
        x = rng.choice(edge) if t < 30 else rng.randrange(p)
        y = rng.choice(edge) if t < 15 else rng.randrange(p)
        em = Emu()
        for i in range(8):
            em.s[mp["a"][i]] = (x >> (32 * i)) & M32
            em.s[mp["b"][i]] = (y >> (32 * i)) & M32
        em.run(E.order)
        got = sum(em.v[mp["o"][i]] << (32 * i) for i in range(8))
        assert got < 2 * p and got % p == x * y * Rinv % p, (hex(p), t)
    return E, mp


def selftest_montmul_lazy(p, trials=300, seed=5):
    """The lazy range used by the BN254 point formulas: for inputs anywhere in [0, 2p) the block's output stays in [0, 2p)
    (needs 4p < 2^256) and is congruent to a*b/R."""
    assert 4 * p < R
    rng = random.Random(seed)
    E, mp = build_montmul(p)
    Rinv = pow(R, -1, p)
    edge = [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p - 2, (1 << 255) % (2 * p)]
    for t in range(trials):
        x = rng.choice(edge) if t < 40 else rng.randrange(2 * p)
        y = rng.choice(edge) if t < 20 else rng.randrange(2 * p)
        em = Emu()
        for i in range(8):
            em.s[mp["a"][i]] = (x >> (32 * i)) & M32
            em.s[mp["b"][i]] = (y >> (32 * i)) & M32
        em.run(E.order)
        got = sum(em.v[mp["o"][i]] << (32 * i) for i in range(8))
        assert got < 2 * p and got % p == x * y * Rinv % p, (hex(p), t, hex(x), hex(y))
    return True


def selftest_finish(p, trials=40, seed=1, lds=False):
    rng = random.Random(seed)
    E, mp = build_beaver_finish(p, key_names=["s%d" % (70 + i) for i in range(8)], lds=lds)
    # substitute the operand placeholders used by i_and / key
    Rinv = pow(R, -1, p)
    edge = [0, 1, p - 1, p - 2, (1 << 255) % p, R % p, (p + 1) // 2]
    for t in range(trials):
        pick = (lambda: rng.choice(edge)) if t < 12 else (lambda: rng.randrange(p))
        vals = {k: pick() for k in ("my_d", "peer_d", "my_e", "peer_e", "b_s", "a_s", "b_m", "a_m", "c_s", "c_m")}
        key = pick()
        for party in (0, 1):
            em = Emu()
            em.mem = vals
            for i in range(8):
                em.s["s%d" % (70 + i)] = (key >> (32 * i)) & M32
            em.s["%[mask]"] = M32 if party == 0 else 0
            em.s["MEMORDER"] = 0
            em.s[S_INV] = (-pow(p, -1, 1 << 32)) & M32
            em.v["v_zero"] = 0
            em.run(E.order)
            d = (vals["my_d"] + vals["peer_d"]) % p
            e = (vals["my_e"] + vals["peer_e"]) % p
            mm = lambda x, y: x * y * Rinv % p          # Montgomery product of Montgomery-form residues
            de = mm(d, e)
            want_s = (mm(d, vals["b_s"]) + mm(e, vals["a_s"]) + vals["c_s"] + (de if party == 0 else 0)) % p
            want_m = (mm(d, vals["b_m"]) + mm(e, vals["a_m"]) + vals["c_m"] + mm(key, de)) % p
            got_s, got_m = em.getv(mp["rs"]), em.getv(mp["rm"])
            assert got_s == want_s, ("share", hex(p), t, party)
            assert got_m == want_m, ("mac", hex(p), t, party)
    return E, mp


def c_string(lines):
    return "\n".join('        "%s\\n\\t"' % ln for ln in lines)


def emit_header(path):
    out = []
    out.append("// GENERATED by tools/gen_asm_kernels.py -- do not edit.  Hand-scheduled gfx950 bodies; see the generator for")
    out.append("// the hazard rules (H1-H4), the register map and the single-lane emulator that validates each stream.")
    out.append("// beaver_finish_asm<F, NT>: NT = 1 adds non-temporal hints on once-streamed data (split-column layout); NT = 2 on the loads only")
    out.append("// (non-temporal 16-byte stores 32 B apart leave the L2 as partial lines: WRITE_SIZE +23..34 % on pure store kernels, probes/write_calib.hip).")
    out.append("#pragma once")
    stats = []
    for fid, (name, p) in enumerate(FIELDS):
        if name in COORD_ONLY:                                 # coordinate fields never carry shares: montmul block only
            continue
        selftest_finish(p, trials=16, seed=fid)               # emulator check (uses placeholder SGPR names for the key)
        out.append("template <> struct HasAsmFinish<%d> { static constexpr bool value = true; };" % fid)
        for nt in (0, 1, 2):
            E, mp = build_beaver_finish(p, nt=nt)              # same stream with the asm operand names %[k0]..%[k7]
            nvalu = sum(1 for i in E.order if i.op in ("mov", "mad", "mul_lo", "addco", "addc", "subco", "subb", "cnd", "and"))
            nmad = sum(1 for i in E.order if i.op == "mad")
            if nt == 0:
                stats.append((name, nvalu, nmad, E.nops, mp["nv"]))
            clob = ['"memory"', '"vcc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mp["first"], mp["nv"])]
            out.append("// %s (NT=%d): %d VALU (%d v_mad_u64_u32), %d H1 wait states, VGPRs v%d..v%d" % (name, nt, nvalu, nmad, E.nops, mp["first"], mp["nv"] - 1))
            out.append("template <> __device__ __forceinline__ void beaver_finish_asm<%d, %d>(u32 off_de, u32 off_col, u32 off_out, const u64* my_d, const u64* my_e," % (fid, nt))
            out.append("        const u64* peer_d, const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s, const u64* b_m, const u64* c_s, const u64* c_m,")
            out.append("        u64* out_s, u64* out_m, const Fe& key, u32 mask) {")
            out.append("    asm volatile(")
            out.append(c_string(E.lines))
            out.append("        :")
            out.append('        : [off_de] "v"(off_de), [off_col] "v"(off_col), [off_out] "v"(off_out), [my_d] "s"(my_d), [my_e] "s"(my_e),')
            out.append('          [peer_d] "s"(peer_d), [peer_e] "s"(peer_e), [a_s] "s"(a_s), [a_m] "s"(a_m), [b_s] "s"(b_s), [b_m] "s"(b_m),')
            out.append('          [c_s] "s"(c_s), [c_m] "s"(c_m), [out_s] "s"(out_s), [out_m] "s"(out_m), [mask] "s"(mask),')
            out.append("          " + ", ".join('[k%d] "s"(key.v[%d])' % (i, i) for i in range(8)))
            out.append("        : " + ", ".join(clob) + ");")
            out.append("}")
    # split layout, result staged through LDS (whole-line stores by the workgroup)
    for fid, (name, p) in enumerate(FIELDS):
        if name in COORD_ONLY:
            continue
        E, mp = build_beaver_finish(p, nt=1, ldsout=True)
        nvalu = sum(1 for i in E.order if i.op in ("mov", "mad", "mul_lo", "addco", "addc", "subco", "subb", "cnd", "and"))
        clob = ['"memory"', '"vcc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mp["first"], mp["nv"])]
        out.append("// %s (split layout, result left in LDS: share at lds_off, MAC at lds_off + 8192): %d VALU, %d H1 wait states" % (name, nvalu, E.nops))
        out.append("template <> __device__ __forceinline__ void beaver_finish_asm_so<%d>(u32 off_de, u32 off_col, u32 lds_off, const u64* my_d, const u64* my_e," % fid)
        out.append("        const u64* peer_d, const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s, const u64* b_m, const u64* c_s, const u64* c_m,")
        out.append("        const Fe& key, u32 mask) {")
        out.append("    asm volatile(")
        out.append(c_string(E.lines))
        out.append("        :")
        out.append('        : [off_de] "v"(off_de), [off_col] "v"(off_col), [lds_off] "v"(lds_off), [my_d] "s"(my_d), [my_e] "s"(my_e),')
        out.append('          [peer_d] "s"(peer_d), [peer_e] "s"(peer_e), [a_s] "s"(a_s), [a_m] "s"(a_m), [b_s] "s"(b_s), [b_m] "s"(b_m),')
        out.append('          [c_s] "s"(c_s), [c_m] "s"(c_m), [mask] "s"(mask),')
        out.append("          " + ", ".join('[k%d] "s"(key.v[%d])' % (i, i) for i in range(8)))
        out.append("        : " + ", ".join(clob) + ");")
        out.append("}")
    # LDS-staged AoS variant of the K2+K3 body
    for fid, (name, p) in enumerate(FIELDS):
        if name in COORD_ONLY:
            continue
        selftest_finish(p, trials=16, seed=fid, lds=True)
        E, mp = build_beaver_finish(p, lds=True)
        nvalu = sum(1 for i in E.order if i.op in ("mov", "mad", "mul_lo", "addco", "addc", "subco", "subb", "cnd", "and"))
        clob = ['"memory"', '"vcc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mp["first"], mp["nv"])]
        out.append("// %s (LDS-staged AoS): %d VALU, %d H1 wait states; a / b / c records read from LDS at lds_off, result written back to the a-region" % (name, nvalu, E.nops))
        out.append("template <> __device__ __forceinline__ void beaver_finish_asm_lds<%d>(u32 off_de, u32 lds_off, const u64* my_d, const u64* my_e," % fid)
        out.append("        const u64* peer_d, const u64* peer_e, const Fe& key, u32 mask) {")
        out.append("    asm volatile(")
        out.append(c_string(E.lines))
        out.append("        :")
        out.append('        : [off_de] "v"(off_de), [lds_off] "v"(lds_off), [my_d] "s"(my_d), [my_e] "s"(my_e), [peer_d] "s"(peer_d), [peer_e] "s"(peer_e),')
        out.append('          [mask] "s"(mask),')
        out.append("          " + ", ".join('[k%d] "s"(key.v[%d])' % (i, i) for i in range(8)))
        out.append("        : " + ", ".join(clob) + ");")
        out.append("}")
    # single Montgomery multiplication blocks
    for fid, (name, p) in enumerate(FIELDS):
        selftest_montmul(p, trials=60, seed=fid)
        E, mp = build_montmul(p)
        nvalu = sum(1 for i in E.order if i.op in ("mov", "mad", "mul_lo", "addco", "addc", "subco", "subb", "cnd", "and"))
        clob = ['"vcc"'] + ['"%s"' % s_ for s_ in MM_CLOBBER_SGPRS] + ['"v%d"' % i for i in mp["used"]]
        out.append("// %s: single Montgomery multiplication, %d VALU, %d H1 wait states; result in [0, 2p)" % (name, nvalu, E.nops))
        out.append("template <> __device__ __forceinline__ Fe fe_mont_mul_asm<%d>(const Fe& a, const Fe& b) {" % fid)
        out.append("    Fe o;")
        out.append("    asm(")
        out.append(c_string(E.lines))
        out.append("        : " + ", ".join('[o%d] "=&v"(o.v[%d])' % (i, i) for i in range(8)))
        out.append("        : " + ", ".join('[a%d] "v"(a.v[%d])' % (i, i) for i in range(8)) + ",")
        out.append("          " + ", ".join('[b%d] "v"(b.v[%d])' % (i, i) for i in range(8)))
        out.append("        : " + ", ".join(clob) + ");")
        out.append("    return o;")
        out.append("}")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "asm_kernels.inc"))
    a = ap.parse_args()
    if a.selftest:
        for name, p in FIELDS:
            try:
                if name in COORD_ONLY:
                    raise AssertionError("coordinate field: no Beaver kernel")
                E, mp = selftest_finish(p, trials=60)
                print("%-14s ok: %d instrs, %d wait states, %d VGPRs" % (name, len(E.order), E.nops, mp["nv"]))
            except AssertionError as ex:
                print("%-14s %s" % (name, ex))
            E, mp = selftest_montmul(p)
            print("%-14s montmul block ok: %d instrs, %d wait states, %d fixed temporaries" % (name, len(E.order), E.nops, len(mp["used"])))
        sys.exit(0)
    for s in emit_header(a.o):
        print("%-14s VALU %d  mad %d  H1 wait states %d  VGPR end %d" % s)
