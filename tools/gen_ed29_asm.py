#!/usr/bin/env python3
"""Generator for the 29-bit-limb form of the Curve25519 (twisted Edwards, a = -1) hand-scheduled kernels (ark-mpc_amd/csrc/ed29_asm_kernels.inc):
window loop, table kernel, fixed-base chain -- the same algorithms, operands and memory formats as tools/gen_ed_asm.py, on NINE unsaturated
29-bit limbs in plain arithmetic modulo q = 2^255 - 19 (no Montgomery form: see gen_ed_asm.py for why the arkworks limbs read as plain
elements are the same projective point).

  * Multiplication: 9 x 29 = 261 bits and 2^261 = 1216 (mod q).  Product scanning into ONE 64-bit column accumulator (the addend of
    v_mad_u64_u32; no carry handling).  The HIGH columns 9..16 go first and leave the limbs h_0..h_8 of floor(a b / 2^261); the low columns
    then take 1216 h_k as one more term, and the carry out of column 8 re-enters limb 0 the same way.  91 multiplier + 41 other instructions
    (squaring 55 + 50) against 72 + 94 (44 + 92) of the 32-bit-limb rows.  The result has nine limbs below 2^29 (limb 1: + 2^15) -- a value
    below 2^261, NOT below 2^255: nothing in the formulas needs more, and the packing step folds 2^255 = 19 when a value leaves for memory.
  * Additions, subtractions, doublings are limb-wise and carry-free (a - b = a + K - b with K a multiple of q whose limbs dominate b's);
    the generator carries bounds (largest limb, largest top limb, largest value) through the bodies and asserts that every multiplier column
    fits 64 bits, as tools/gen_ec29_asm.py does for BN254.
  * The group law is complete (add-2008-hwcd-3, a = -1): no flags.  A negative digit swaps (Y+X, Y-X) and replaces 2dT by K - 2dT; the
    identity entry needs no special case (K - 0 is a multiple of q).
  * Memory formats are the 32-bit kernels': 8 x 32-bit words per coordinate.  A value is packed after one fold of bit 255 upwards (x 19
    into limb 0) and a strict carry pass, so every stored word string is below 2^256.

Bodies are executed by the single-lane emulator against the affine Edwards law in Python integers (--selftest, tests/test_asm_generator.py).

Reference semantics: CurvePoint * Scalar on ark_curve25519::EdwardsProjective (online-phase/src/algebra/curve/curve.rs:403-409), PointShare *
Scalar (curve/share.rs:108-114), batch_mul_generator (authenticated_curve.rs:754-780).
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_asm_kernels as G
import gen_ec_asm as EC
import gen_ec29_asm as E29
import gen_ed_asm as ED
from gen_asm_kernels import Ins, Emitter, M32, regs_of
from gen_ec29_asm import FV, NL, M29, limbs29, val29, i_mad, i_and, i_shr64, i_mov, i_add, i_sub, i_shl, i_lshr, i_cnd, seq_unpack, seq_pack, mem_ops

Q = ED.Q
D_ED = ED.D_ED
FOLD = 1216                                   # 2^261 mod q
assert (1 << 261) % Q == FOLD
N_WINDOWS, N_TABLE = ED.N_WINDOWS, ED.N_TABLE
GEN_C, GEN_WINDOWS, GEN_ENTRIES = ED.GEN_C, ED.GEN_WINDOWS, ED.GEN_ENTRIES

S_JUNK, S_FOLD, S_MASK = E29.S_JUNK, "s18", "s19"
S_C = E29.S_C                                 # a constant operand (2d)
S_STEP, S_N4, S_N128, S_TMP, S_DBL = "s38", "s39", "s40", "s41", "s42"
S_NEG = "s[46:47]"
CLOBBER_SGPRS = ["s%d" % i for i in range(16, 58)]

MUL_LMAX = M29 + (1 << 15)                    # limb 1 of a product takes the last carry
MUL_VMAX = (1 << 261) + (1 << 46)


def i_mad24(d, a, b, c): return Ins("v_mad_u32_u24 %s, %s, %s, %s" % (d, a, G.src(b), c), "mad24", (d, a, b, c), rd=regs_of(a, b, c), wr=[d])


class Emu(E29.Emu29):
    def run(self, order):
        for ins in order:
            if ins.op == "mad24":
                d, a, b, c = ins.args
                x, y = self.rd(a), self.rd(b)
                assert x < (1 << 24) and y < (1 << 24), "24-bit multiplier operand"
                r = x * y + self.rd(c)
                assert r <= M32
                self.v[d] = r
            else:
                E29.Emu29.run(self, [ins])


def fv_mul(r):
    return FV(r, MUL_LMAX, M29, MUL_VMAX)


_K_cache = {}


def k_for(b):
    """limbs of a multiple of q that dominate every limb of b (see gen_ec29_asm.k_for)"""
    u = -(-b.lmax // M29)
    key = (u, b.tmax)
    if key not in _K_cache:
        c = max(1, ((b.tmax + u) << 232) // Q)
        while True:
            l = limbs29(c * Q)
            if l[8] - u >= b.tmax:
                break
            c += 1
        k = [l[0] + (u << 29)] + [l[i] - u + (u << 29) for i in range(1, 8)] + [l[8] - u]
        assert val29(k) == c * Q and min(k) >= 0 and all(k[i] >= b.lmax for i in range(8)) and max(k) <= M32
        _K_cache[key] = k
    return _K_cache[key]


class Bld:
    def __init__(self, rm):
        self.rm, self.seq, self.sigs = rm, [], []

    def mul(self, a, b, out):
        """out = a b (mod q), nine limbs (<= 2^29 - 1; limb 1 <= + 2^15); `out` must not overlap an operand.  a is b: squaring."""
        rm = self.rm
        acc, acc2, h, W = rm.acc, rm.acc2, rm.h, rm.W
        sq = a is b
        assert not (set(out) & (set(a.r) | set(b.r))), "product scanning reads every operand limb until the last column"
        al, bl = [a.lmax] * 8 + [a.tmax], [b.lmax] * 8 + [b.tmax]
        col = max(sum(al[i] * bl[k - i] for i in range(NL) if 0 <= k - i < NL) for k in range(2 * NL - 1)) + FOLD * M32 + (1 << 36)
        assert col < (1 << 64), "multiplier column can overflow: normalise an operand (%d bits)" % col.bit_length()
        assert (a.vmax * b.vmax) >> (261 + 232) < (1 << 32), "high part's top limb must fit 32 bits"
        self.sigs.append((a.lmax, a.tmax, b.lmax, b.tmax, sq))
        if sq:
            assert 2 * a.big <= M32
            self.seq += [i_shl(W[j], a.r[j], 1) for j in range(NL)]

        def terms(k):
            t = []
            for i in range(NL):
                j = k - i
                if not 0 <= j < NL:
                    continue
                if sq:
                    if i < j: t.append((a.r[i], W[j]))
                    elif i == j: t.append((a.r[i], a.r[i]))
                else:
                    t.append((a.r[i], b.r[j]))
            return t
        first = True
        for k in range(NL, 2 * NL - 1):                     # high columns: h = floor(a b / 2^261) as limbs
            for x, y in terms(k):
                self.seq.append(i_mad(acc, x, y, 0 if first else acc))
                first = False
            self.seq += [i_and(h[k - NL], S_MASK, acc[0]), i_shr64(acc, acc, 29)]
        self.seq.append(i_mov(h[NL - 1], acc[0]))
        first = True
        for k in range(NL):                                 # low columns + 1216 h_k
            for x, y in terms(k) + [(h[k], S_FOLD)]:
                self.seq.append(i_mad(acc, x, y, 0 if first else acc))
                first = False
            if k == 0:
                self.seq += [i_and(acc2[0], S_MASK, acc[0]), i_mov(acc2[1], 0)]
            else:
                self.seq.append(i_and(out[k], S_MASK, acc[0]))
            self.seq.append(i_shr64(acc, acc, 29))
        # the carry out of column 8 (weight 2^261 = 1216) re-enters limb 0; its own carry goes to limb 1
        self.seq += [i_mad(acc2, acc[0], S_FOLD, acc2), i_mad24(acc2[1], acc[1], S_FOLD, acc2[1]),
                     i_and(out[0], S_MASK, acc2[0]), i_shr64(acc2, acc2, 29), i_add(out[1], acc2[0], out[1])]
        return fv_mul(out)

    def add(self, a, b, out):
        self.seq += [i_add(out[j], a.r[j], b.r[j]) for j in range(NL)]
        return FV(out, a.lmax + b.lmax, a.tmax + b.tmax, a.vmax + b.vmax)

    def shl(self, a, sh, out):
        self.seq += [i_shl(out[j], a.r[j], sh) for j in range(NL)]
        return FV(out, a.lmax << sh, a.tmax << sh, a.vmax << sh)

    def sub(self, a, b, out):
        k = k_for(b)
        for j in range(NL):
            self.seq += [i_sub(out[j], a.r[j], b.r[j]), i_add(out[j], k[j], out[j])]
        return FV(out, a.lmax + max(k[:8]), a.tmax + k[8], a.vmax + val29(k))

    def neg(self, b, out):
        k = k_for(b)
        self.seq += [i_sub(out[j], k[j], b.r[j]) for j in range(NL)]
        return FV(out, max(k[:8]), k[8], val29(k))

    def norm(self, a, out=None):
        """fold the top limb's bits from 23 upwards into limb 0 (2^255 = 19), then one parallel carry pass: limbs <= 2^29 - 1 + a few, top limb
        below 2^23 + a few, value below 2^255 + 2^233"""
        out = out or a.r
        W = self.rm.W
        t = self.rm.h[0]
        hi = a.tmax >> 23
        assert hi < (1 << 24) and a.lmax + 19 * hi <= M32
        self.seq += [i_lshr(t, a.r[8], 23), i_mad24(out[0], t, 19, a.r[0])]
        l0 = a.lmax + 19 * hi
        self.seq += [i_lshr(W[0], out[0], 29)] + [i_lshr(W[j], a.r[j], 29) for j in range(1, 8)]
        self.seq += [i_and(out[0], S_MASK, out[0])]
        for j in range(1, 8):
            self.seq += [i_and(out[j], S_MASK, a.r[j]), i_add(out[j], W[j - 1], out[j])]
        self.seq += [i_and(out[8], (1 << 23) - 1, a.r[8]), i_add(out[8], W[7], out[8])]
        c = max(l0, a.lmax) >> 29
        return FV(out, M29 + c, (1 << 23) - 1 + c, ((1 << 23) + c) << 232)

    def movs(self, a, out):
        self.seq += [i_mov(out[j], a.r[j]) for j in range(NL)]
        return FV(out, a.lmax, a.tmax, a.vmax)

    def pack(self, a, words):
        """a (in place) -> 8 words below 2^256: fold the top limb's bits from 23 upwards (2^255 = 19) into limb 0, strict serial carry pass, pack"""
        t = self.rm.W[0]
        r = a.r
        assert (a.tmax >> 23) < (1 << 24) and a.lmax + 19 * (a.tmax >> 23) <= M32
        self.seq += [i_lshr(t, r[8], 23), i_and(r[8], (1 << 23) - 1, r[8]), i_mad24(r[0], t, 19, r[0])]
        c = self.rm.W[1]
        for j in range(8):
            self.seq += [i_lshr(c, r[j], 29), i_and(r[j], S_MASK, r[j]), i_add(r[j + 1], c, r[j + 1])]
        assert ((1 << 23) - 1 + ((a.lmax + 19 * (a.tmax >> 23)) >> 29) + 8) < (1 << 24)
        self.seq += seq_pack(r, words)


# ---- bodies ------------------------------------------------------------------------------------------------------------------------
def acc_fvs(rm):
    return fv_mul(rm.X1), fv_mul(rm.Y1), fv_mul(rm.Z1), fv_mul(rm.T1)


def seq_double(rm, with_t, B=None):
    """dbl-2008-hwcd (a = -1) without negations, as tools/gen_ed_asm.py: A = X^2, B = Y^2, Cc = 2 Z^2, E = 2 X Y, G = B - A, F' = Cc - G,
    Hn = A + B; (X3, Y3, Z3, T3) = (E F', G Hn, F' G, E Hn).  Scratch: the entry registers, A / Bv / Cv, W."""
    B = B or Bld(rm)
    X, Y, Z, T = acc_fvs(rm)
    A = B.mul(X, X, rm.A)
    Bq = B.mul(Y, Y, rm.Bv)
    E = B.shl(B.mul(X, Y, rm.QM), 1, rm.QM)
    Cc = B.shl(B.mul(Z, Z, rm.QP), 1, rm.QP)
    Gv = B.norm(B.sub(Bq, A, rm.QT))
    Hn = B.add(A, Bq, rm.Cv)
    Fp = B.norm(B.sub(Cc, Gv, rm.QZ))
    B.mul(E, Fp, rm.X1)
    B.mul(Gv, Hn, rm.Y1)
    B.mul(Fp, Gv, rm.Z1)
    if with_t:
        B.mul(E, Hn, rm.T1)
    return B


def entry_fvs(rm, qt=None):
    n = lambda r: FV(r, M29, (1 << 24) - 1, (1 << 256) - 1)          # unpacked from 8 words
    return n(rm.QP), n(rm.QM), qt or n(rm.QT), n(rm.QZ)


def seq_add(rm, with_t, B=None, qt=None, niels=False, qpm=None):
    """add-2008-hwcd-3 (a = -1) with a cached second operand (Y2+X2, Y2-X2, 2d T2, 2 Z2) in QP, QM, QT, QZ; niels: the operand is affine
    (Z2 = 1: D = 2 Z1 is a limb shift).  Complete on this curve.  qt: bounds of the 2dT operand (the loop negates it conditionally)."""
    B = B or Bld(rm)
    X, Y, Z, T = acc_fvs(rm)
    QP, QM, QT, QZ = entry_fvs(rm, qt)
    if qpm is not None:                                     # (y + x, y - x) formed in registers (the MSM fold): wider bounds than unpacked words
        QP, QM = FV(rm.QP, qpm.lmax, qpm.tmax, qpm.vmax), FV(rm.QM, qpm.lmax, qpm.tmax, qpm.vmax)
    s1 = B.sub(Y, X, rm.A)
    s2 = B.add(Y, X, rm.Bv)
    PA = B.mul(s1, QM, rm.Cv)
    PB = B.mul(s2, QP, rm.A)
    PC = B.mul(T, QT, rm.Bv)
    PD = B.shl(Z, 1, rm.QM) if niels else B.mul(Z, QZ, rm.QM)
    E = B.norm(B.sub(PB, PA, rm.QP))
    H = B.add(PB, PA, rm.QT)
    F = B.norm(B.sub(PD, PC, rm.QZ))
    Gv = B.add(PD, PC, rm.A)
    B.mul(E, F, rm.X1)
    B.mul(Gv, H, rm.Y1)
    B.mul(F, Gv, rm.Z1)
    if with_t:
        B.mul(E, H, rm.T1)
    B.EH = (E, H)
    return B


def seq_cached(rm, d2):
    """(A, Bv, Cv, QZ) = (Y1 + X1, Y1 - X1, 2d T1, 2 Z1) of the accumulator; returns the builder and the four values"""
    B = Bld(rm)
    X, Y, Z, T = acc_fvs(rm)
    return B, (B.add(Y, X, rm.A), B.sub(Y, X, rm.Bv), B.mul(T, d2, rm.Cv), B.shl(Z, 1, rm.QZ))


# ---- registers -----------------------------------------------------------------------------------------------------------------------
class RegMap:
    """A 9-limb vector is registers 1..9 of an even-aligned run of ten: limbs 1..8 start on an even register, so the eight 32-bit words of a
    coordinate can be loaded straight into them (dwordx4 pairs) and unpacked IN PLACE (seq_unpack_inplace) -- no landing registers."""

    def __init__(self, first=8):
        rg = G.Regs(first)
        v9 = lambda: rg.vec(10, 2)[1:]
        self.X1, self.Y1, self.Z1, self.T1 = v9(), v9(), v9(), v9()
        self.QP, self.QM, self.QT, self.QZ = v9(), v9(), v9(), v9()
        self.A, self.Bv, self.Cv = v9(), v9(), v9()
        self.W = v9()
        self.h = rg.vec(9)
        self.acc, self.acc2 = rg.pair(), rg.pair()
        self.ST = rg.vec(8, 2)                                                # one coordinate's words on their way to memory
        self.LD = rg.vec(8, 2)                                                # with ST: landing registers of the first half of a prefetched table entry
        self.rec, self.off, self.tid4, self.tid128, self.tmp = (rg.one() for _ in range(5))
        self.first, self.end = first, rg.next
        assert self.end <= 168, self.end


def seq_unpack_inplace(q):
    """the eight words sit in q[1..8]; afterwards q[0..8] are the nine limbs.  Limb i needs words i - 1 and i (29 i / 32 = i - 1 for i <= 8), i.e.
    registers i and i + 1, and is written to register i: ascending order is safe, and the scheduler keeps it (register dependences)."""
    s = [i_and(q[0], S_MASK, q[1])]
    for i in range(1, 8):
        sh = (29 * i) & 31
        assert (29 * i) >> 5 == i - 1
        s += [E29.i_alignbit(q[i], q[i + 1], q[i], sh), i_and(q[i], S_MASK, q[i])]
    s += [i_lshr(q[8], q[8], 8)]
    return s


def prologue(A):
    A("s_nop 1")
    A("s_mov_b32 %s, %d" % (S_FOLD, FOLD))
    A("s_mov_b32 %s, 0x%08x" % (S_MASK, M29))


def const_to(A, value):
    for j, l in enumerate(limbs29(value)):
        A("s_mov_b32 %s, 0x%08x" % (S_C[j], l))
    l = limbs29(value)
    return FV(S_C, max(l[:8]), l[8], value)


def emu_for():
    em = Emu()
    em.s[S_FOLD] = FOLD
    em.s[S_MASK] = M29
    return em


# ---- self-test -------------------------------------------------------------------------------------------------------------------------
def _rep(rng, v, extreme=False):
    """nine limbs of a representative of v (mod q) as a product leaves it: value below 2^261, limb 1 may carry 2^15 extra"""
    cmax = ((1 << 261) - 1 - v) // Q
    val = v + (cmax if extreme else rng.randrange(cmax + 1)) * Q
    l = limbs29(val)
    if l[2] > 0 and (extreme or rng.random() < 0.3):
        l[2] -= 1; l[1] += 1 << 29
        if l[1] > MUL_LMAX:
            l[2] += 1; l[1] -= 1 << 29
    return l


def selftest(trials=30, seed=13):
    rng = random.Random(seed)
    out = {}
    ext = lambda p_, z_: (p_[0] * z_ % Q, p_[1] * z_ % Q, z_, p_[0] * p_[1] * z_ % Q)

    def run(E, rm, accv, entry=None, extreme=False):
        em = emu_for()
        for regs, v in zip((rm.X1, rm.Y1, rm.Z1, rm.T1), accv):
            em.set9(regs, _rep(rng, v, extreme))
        if entry:
            for regs, v in zip((rm.QP, rm.QM, rm.QT, rm.QZ), entry):
                em.set9(regs, limbs29(v))
        em.run(E.order)
        got = [[em.v[r] for r in regs] for regs in (rm.X1, rm.Y1, rm.Z1, rm.T1)]
        for l in got:
            assert max(l[0], *l[2:]) <= M29 and l[1] <= MUL_LMAX
        return [val29(l) % Q for l in got]

    for with_t in (False, True):
        rm = RegMap()
        Ed = Emitter(); Ed.schedule(seq_double(rm, with_t).seq)
        Ea = Emitter(); Ea.schedule(seq_add(rm, with_t).seq)
        out[with_t] = (Ed, Ea)
        for t in range(trials):
            P = ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)) if t % 7 else (0, 1)
            z = rng.randrange(1, Q)
            accv = ext(P, z)
            gx, gy, gz, gt = run(Ed, rm, accv, extreme=t % 5 == 4)
            zi = pow(gz, -1, Q)
            want = ED.ed_add_aff(P, P)
            assert (gx * zi % Q, gy * zi % Q) == want, ("double", with_t, t)
            if with_t:
                assert gt * zi % Q == want[0] * want[1] % Q
            kind = t % 6
            Qp = (0, 1) if kind == 0 else (P if kind == 1 else ((Q - P[0]) % Q, P[1]) if kind == 2 else ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)))
            z2 = rng.randrange(1, Q)
            X2, Y2, Z2, T2 = ext(Qp, z2)
            big = lambda v: v + Q if (t % 3 == 0 and v + Q < (1 << 256)) else v
            entry = (big((Y2 + X2) % Q), big((Y2 - X2) % Q), big(2 * D_ED * T2 % Q), big(2 * Z2 % Q))
            gx, gy, gz, gt = run(Ea, rm, accv, entry, extreme=t % 5 == 4)
            zi = pow(gz, -1, Q)
            want = ED.ed_add_aff(P, Qp)
            assert (gx * zi % Q, gy * zi % Q) == want, ("add", with_t, t, kind)
            if with_t:
                assert gt * zi % Q == want[0] * want[1] % Q
    rm = RegMap()
    En = Emitter(); En.schedule(seq_add(rm, True, niels=True).seq)
    for t in range(trials):
        P = ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)) if t % 7 else (0, 1)
        z = rng.randrange(1, Q)
        kind = t % 5
        Qp = (0, 1) if kind == 0 else (P if kind == 1 else ((Q - P[0]) % Q, P[1]) if kind == 2 else ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)))
        entry = ((Qp[1] + Qp[0]) % Q, (Qp[1] - Qp[0]) % Q, 2 * D_ED * Qp[0] * Qp[1] % Q, 0)
        gx, gy, gz, gt = run(En, rm, ext(P, z), entry)
        zi = pow(gz, -1, Q)
        want = ED.ed_add_aff(P, Qp)
        assert (gx * zi % Q, gy * zi % Q) == want and gt * zi % Q == want[0] * want[1] % Q, ("niels", t, kind)
    # cached form + packing: words below 2^256, right residues; in-place unpacking gives the same value back
    rm = RegMap()
    d2 = 2 * D_ED % Q
    l2 = limbs29(d2)
    Bc, ents = seq_cached(rm, FV(S_C, max(l2[:8]), l2[8], d2))
    Ec = Emitter(); Ec.schedule(Bc.seq)
    packs = []
    for e in ents:
        Bp = Bld(rm); Bp.pack(e, rm.ST)
        Ep = Emitter(); Ep.schedule(Bp.seq); packs.append(Ep)
    Eu = Emitter(); Eu.schedule(seq_unpack_inplace(rm.X1))
    for t in range(12):
        vals = [rng.randrange(Q) for _ in range(4)]
        em = emu_for()
        for j, l in enumerate(l2):
            em.s[S_C[j]] = l
        for regs, v in zip((rm.X1, rm.Y1, rm.Z1, rm.T1), vals):
            em.set9(regs, _rep(rng, v, t % 3 == 0))
        em.run(Ec.order)
        x, y, z, t_ = vals
        for Ep, want in zip(packs, (y + x, y - x, d2 * t_, 2 * z)):
            em.run(Ep.order)
            got = sum(em.v[r] << (32 * i) for i, r in enumerate(rm.ST))
            assert got < (1 << 256) and got % Q == want % Q
            for i, r in enumerate(rm.ST):
                em.v[rm.X1[1 + i]] = em.v[r]
            em.run(Eu.order)
            assert em.get9(rm.X1) == got and max(em.v[r] for r in rm.X1[:8]) <= M29
    return out


def selftest_extremes():
    """every multiplication of the bodies with all limbs of both operands at the bounds carried to it"""
    rm = RegMap()
    sigs = set()
    for b in (seq_double(rm, True), seq_add(rm, True), seq_add(rm, True, niels=True)):
        sigs |= set(b.sigs)
    Bn = Bld(rm)
    qn = Bn.neg(FV(rm.QT, M29, (1 << 24) - 1, (1 << 256) - 1), rm.A)
    sigs |= set(seq_add(rm, True, qt=FV(rm.QT, qn.lmax, qn.tmax, qn.vmax)).sigs)
    d2v = 2 * D_ED % Q
    l2 = limbs29(d2v)
    Bm, qp, qt_in = seq_member(rm, FV(S_C, max(l2[:8]), l2[8], d2v))
    _, qt = seq_member_select(rm, qt_in, "s[44:45]")
    sigs |= set(Bm.sigs) | set(seq_add(rm, True, qt=qt, niels=True, qpm=qp).sigs)
    for (al, at, bl, bt, sq) in sorted(sigs):
        B = Bld(rm)
        la, lb = [al] * 8 + [at], [bl] * 8 + [bt]
        # the value bound of the real operands is far below what all-maximal limbs represent: clamp the top limbs so that h_8 fits (as in the bodies)
        la[8] = min(la[8], (1 << 31) - 1); lb[8] = min(lb[8], (1 << 31) - 1)
        while (val29(la) * val29(la if sq else lb)) >> (261 + 232) >= (1 << 31):
            la[8] >>= 1; lb[8] >>= 1
        a = FV(rm.X1, al, la[8], val29(la))
        em = emu_for(); em.set9(rm.X1, la)
        if sq:
            B.mul(a, a, rm.Z1); want = val29(la) ** 2
        else:
            b = FV(rm.Y1, bl, lb[8], val29(lb)); em.set9(rm.Y1, lb)
            B.mul(a, b, rm.Z1); want = val29(la) * val29(lb)
        E = Emitter(); E.schedule(B.seq)
        em.run(E.order)
        got = [em.v[r] for r in rm.Z1]
        assert val29(got) % Q == want % Q and max(got[0], *got[2:]) <= M29 and got[1] <= MUL_LMAX, (al, at, bl, bt, sq)
    return len(sigs)


# ---- the loop ------------------------------------------------------------------------------------------------------------------------------
def _sched(L, seq, pre=None):
    E = Emitter()
    if pre: E.lastw.update(pre)
    E.schedule(seq)
    L.extend(E.lines)
    return E


def _neg_select(rm, L):
    """negative digit: -(x, y) has the cached / Niels form (Y-X, Y+X, K - 2dT, 2Z).  Returns the bounds of QT afterwards."""
    Bn = Bld(rm)
    qt_in = FV(rm.QT, M29, (1 << 24) - 1, (1 << 256) - 1)
    qn = Bn.neg(qt_in, rm.A)
    Bn.seq += [i_mov(rm.Bv[j], rm.QP[j]) for j in range(NL)]
    Bn.seq += [i_cnd(rm.QP[j], rm.QP[j], rm.QM[j], S_NEG) for j in range(NL)]
    Bn.seq += [i_cnd(rm.QM[j], rm.QM[j], rm.Bv[j], S_NEG) for j in range(NL)]
    Bn.seq += [i_cnd(rm.QT[j], rm.QT[j], rm.A[j], S_NEG) for j in range(NL)]
    _sched(L, Bn.seq, pre={S_NEG: -1})
    return FV(rm.QT, max(qt_in.lmax, qn.lmax), max(qt_in.tmax, qn.tmax), max(qt_in.vmax, qn.vmax))


def _store_vals(rm, L, A, vals, base, off):
    """four values packed one after the other through the staging words, 32 bytes each at off"""
    ld, st = mem_ops(A)
    for k, e in enumerate(vals):
        Bp = Bld(rm); Bp.pack(e, rm.ST)
        _sched(L, Bp.seq)
        st(rm.ST, off, base, 32 * k)


def _load_vecs(rm, A, vecs, base, off, nbytes=32):
    ld, st = mem_ops(A)
    for k, q in enumerate(vecs):
        ld(q[1:], off, base, nbytes * k)


def emit_loop():
    """Operands as ed_smul_loop_asm (tools/gen_ed_asm.py): %[tid] %[ptid] (VGPR), %[n] %[np] (SGPR), %[tab] %[dig] %[res] (SGPR pairs)."""
    rm = RegMap()
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s: "%s_%%=" % s
    prologue(A)
    A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
    A("v_lshlrev_b32_e32 %s, 7, %%[ptid]" % rm.tid128)
    A("s_lshl_b32 %s, %%[n], 2" % S_N4)
    A("s_lshl_b32 %s, %%[np], 7" % S_N128)
    for j in range(NL):                                                   # accumulator = identity (0, 1, 1, 0)
        A("v_mov_b32_e32 %s, 0" % rm.X1[j])
        A("v_mov_b32_e32 %s, %d" % (rm.Y1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, %d" % (rm.Z1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, 0" % rm.T1[j])
    A("s_mov_b32 %s, 0" % S_STEP)
    EC.align_head(A)
    A(lbl("E_step") + ":")
    A("s_mul_i32 %s, %s, %s" % (S_TMP, S_STEP, S_N4))
    A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid4))
    A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.off))
    A("s_cmp_eq_u32 %s, 0" % S_STEP)
    A("s_cbranch_scc1 " + lbl("E_pref"))
    A("s_mov_b32 %s, 4" % S_DBL)                                          # four doublings without T, the fifth with it
    EC.align_head(A)
    A(lbl("E_dbl") + ":")
    Ed = _sched(L, seq_double(rm, False).seq)
    A("s_sub_u32 %s, %s, 1" % (S_DBL, S_DBL))
    A("s_cmp_lg_u32 %s, 0" % S_DBL)
    A("s_cbranch_scc1 " + lbl("E_dbl"))
    # the first half of this step's table entry (Y+X, Y-X: 64 of its 128 bytes, one cache line) is requested before the fifth doubling, into
    # registers no body touches; the second half is read after it and finds the line in the cache
    A(lbl("E_pref") + ":")
    A("s_waitcnt vmcnt(0)")
    A("v_and_b32_e32 %s, 31, %s" % (rm.tmp, rm.rec))
    A("v_mul_lo_u32 %s, %s, %s" % (rm.tmp, rm.tmp, S_N128))
    A("v_add_u32_e32 %s, %s, %s" % (rm.off, rm.tmp, rm.tid128))
    ld(rm.ST, rm.off, "tab", 0); ld(rm.LD, rm.off, "tab", 32)
    A("s_cmp_eq_u32 %s, 0" % S_STEP)
    A("s_cbranch_scc1 " + lbl("E_have"))
    Edt = _sched(L, seq_double(rm, True).seq)
    A(lbl("E_have") + ":")
    ld(rm.QT[1:], rm.off, "tab", 64); ld(rm.QZ[1:], rm.off, "tab", 96)
    A("v_and_b32_e32 %s, 32, %s" % (rm.tmp, rm.rec))
    A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NEG, rm.tmp))
    A("s_waitcnt vmcnt(0)")
    _sched(L, seq_unpack(rm.ST, rm.QP) + seq_unpack(rm.LD, rm.QM) + seq_unpack_inplace(rm.QT) + seq_unpack_inplace(rm.QZ))
    qt = _neg_select(rm, L)
    Ba = seq_add(rm, False, qt=qt)
    Ea = _sched(L, Ba.seq)
    # the very last addition also produces T (the result is stored in extended coordinates): T3 = E H as seq_add leaves them
    A("s_cmp_lg_u32 %s, %d" % (S_STEP, N_WINDOWS - 1))
    A("s_cbranch_scc1 " + lbl("E_add_not"))
    Bt = Bld(rm); Bt.mul(Ba.EH[0], Ba.EH[1], rm.T1)
    _sched(L, Bt.seq)
    A(lbl("E_add_not") + ":")
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_lt_u32 %s, %d" % (S_STEP, N_WINDOWS))
    A("s_cbranch_scc1 " + lbl("E_step"))
    A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
    _store_vals(rm, L, A, (fv_mul(rm.X1), fv_mul(rm.Y1), fv_mul(rm.T1), fv_mul(rm.Z1)), "res", rm.tid128)     # ark-ec order: x, y, t, z
    A("s_waitcnt vmcnt(0)")
    mc = lambda seq: sum(1 for i in seq if i.op in ("mad", "mad24"))
    st_ = dict(double=len(Ed.order), double_t=len(Edt.order), add=len(Ea.order), vgpr_end=rm.end,
               loop_mults=(N_WINDOWS - 1) * (4 * mc(Ed.order) + mc(Edt.order)) + N_WINDOWS * mc(Ea.order) + mc(Bt.seq))
    return L, rm, st_


def emit_table():
    """Operands as ed_smul_table_asm: %[tid] %[poff] (VGPR), %[n] (SGPR), %[pts] %[tab] (SGPR pairs).  Cached entries (Y+X, Y-X, 2dT, 2Z) of
    0*P .. 16*P, 128 bytes each, any projective representative."""
    rm = RegMap()
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s_: "%s_%%=" % s_
    QV = (rm.QP, rm.QM, rm.QT, rm.QZ)
    prologue(A)
    d2 = const_to(A, 2 * D_ED % Q)
    A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
    A("s_lshl_b32 %s, %%[n], 7" % S_N128)
    _load_vecs(rm, A, (rm.X1, rm.Y1, rm.T1, rm.Z1), "pts", "%[poff]")     # ark-ec order x, y, t, z

    def entry_off(index):
        if isinstance(index, int):
            A("s_mul_i32 %s, %s, %d" % (S_TMP, S_N128, index))
        else:
            A("s_mul_i32 %s, %s, %s" % (S_TMP, S_N128, index))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid128))

    def cached(index):
        Bc, ents = seq_cached(rm, d2)
        E = _sched(L, Bc.seq)
        entry_off(index)
        _store_vals(rm, L, A, ents, "tab", rm.off)
        return E

    # entry 0: the identity (0 : 1 : 1 : 0) -> (1, 1, 0, 2)
    for k, v in enumerate((1, 1, 0, 2)):
        for j in range(8):
            A("v_mov_b32_e32 %s, %d" % (rm.ST[j], v if j == 0 else 0))
        st(rm.ST, rm.tid128, "tab", 32 * k)
    A("s_waitcnt vmcnt(0)")
    _sched(L, sum((seq_unpack_inplace(q) for q in (rm.X1, rm.Y1, rm.T1, rm.Z1)), []))       # any 256-bit strings: inside a product's bounds
    Ec = cached(1)
    Ed = _sched(L, seq_double(rm, True).seq)
    cached(2)
    A("s_mov_b32 %s, 3" % S_STEP)
    A(lbl("T_next") + ":")
    A("s_waitcnt vmcnt(0)")                                                # entry 1 has landed
    A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_N128, rm.tid128))
    _load_vecs(rm, A, QV, "tab", rm.off)
    A("s_waitcnt vmcnt(0)")
    _sched(L, sum((seq_unpack_inplace(q) for q in QV), []))
    Ea = _sched(L, seq_add(rm, True).seq)
    cached(S_STEP)
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_le_u32 %s, 16" % S_STEP)
    A("s_cbranch_scc1 " + lbl("T_next"))
    A("s_waitcnt vmcnt(0)")
    mc = lambda E: sum(1 for i in E.order if i.op in ("mad", "mad24"))
    return L, rm, dict(table_mults=16 * mc(Ec) + mc(Ed) + 14 * mc(Ea), vgpr_end=rm.end)


def emit_gen_chain():
    """Operands as ed_gen_chain_asm: %[tid] (VGPR), %[n] (SGPR), %[dig] %[tab] %[res] (SGPR pairs): 23 additions of tabulated affine multiples
    (plain Niels entries of 96 bytes; index 0 = the identity), no doublings."""
    rm = RegMap()
    L = []
    A = L.append
    lbl = lambda s_: "%s_%%=" % s_
    QV = (rm.QP, rm.QM, rm.QT)
    prologue(A)
    A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
    A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
    A("s_lshl_b32 %s, %%[n], 2" % S_N4)
    for j in range(NL):
        A("v_mov_b32_e32 %s, 0" % rm.X1[j])
        A("v_mov_b32_e32 %s, %d" % (rm.Y1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, %d" % (rm.Z1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, 0" % rm.T1[j])
    A("s_mov_b32 %s, 0" % S_STEP)
    A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.tid4))
    EC.align_head(A)
    A(lbl("G_step") + ":")
    A("s_waitcnt vmcnt(0)")
    A("v_and_b32_e32 %s, 0x7fffffff, %s" % (rm.tmp, rm.rec))
    A("v_mul_u32_u24_e32 %s, 96, %s" % (rm.off, rm.tmp))
    _load_vecs(rm, A, QV, "tab", rm.off)
    A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
    # the next window's record travels while this addition runs
    A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
    A("s_min_u32 %s, %s, %d" % (S_TMP, S_TMP, GEN_WINDOWS - 1))
    A("s_mul_i32 %s, %s, %s" % (S_TMP, S_TMP, S_N4))
    A("v_add_u32_e32 %s, %s, %s" % (rm.tmp, S_TMP, rm.tid4))
    A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.tmp))
    A("s_waitcnt vmcnt(1)")
    _sched(L, sum((seq_unpack_inplace(q) for q in QV), []))
    qt = _neg_select(rm, L)
    Ea = _sched(L, seq_add(rm, True, qt=qt, niels=True).seq)
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_lt_u32 %s, %d" % (S_STEP, GEN_WINDOWS))
    A("s_cbranch_scc1 " + lbl("G_step"))
    A("s_waitcnt vmcnt(0)")
    _store_vals(rm, L, A, (fv_mul(rm.X1), fv_mul(rm.Y1), fv_mul(rm.T1), fv_mul(rm.Z1)), "res", rm.tid128)
    A("s_waitcnt vmcnt(0)")
    mult = sum(1 for i in Ea.order if i.op in ("mad", "mad24"))
    return L, rm, dict(add=len(Ea.order), chain_mults=GEN_WINDOWS * mult, vgpr_end=rm.end)


def seq_member(rm, d2, unpack=False):
    """a plain affine member (x, y) -- eight words each in ST / LD when `unpack`, else nine limbs in A / Bv -- to its Niels form
    (QP, QM, QT) = (y + x, y - x, 2d x y); QP and QM both normalised (a negative digit swaps them).  Returns (builder, bounds of QP / QM, bounds of QT)."""
    B = Bld(rm)
    if unpack:
        B.seq += seq_unpack(rm.ST, rm.A) + seq_unpack(rm.LD, rm.Bv)
    n = lambda r: FV(r, M29, (1 << 24) - 1, (1 << 256) - 1)
    x, y = n(rm.A), n(rm.Bv)
    qp_ = B.norm(B.add(y, x, rm.QP))
    qm_ = B.norm(B.sub(y, x, rm.QM))
    xy = B.mul(x, y, rm.Cv)
    qt = B.mul(xy, d2, rm.QT)
    return B, FV(rm.QP, max(qp_.lmax, qm_.lmax), max(qp_.tmax, qm_.tmax), max(qp_.vmax, qm_.vmax)), qt


def seq_member_select(rm, qt_in, S_NZ):
    """negative digit (mask S_NEG): (QP, QM, QT) -> (QM, QP, K - QT); lane past the end of its task (mask S_NZ clear): the identity (1, 1, 0)"""
    B = Bld(rm)
    qn = B.neg(qt_in, rm.A)
    B.seq += [i_mov(rm.Bv[j], rm.QP[j]) for j in range(NL)]
    B.seq += [i_cnd(rm.QP[j], rm.QP[j], rm.QM[j], S_NEG) for j in range(NL)]
    B.seq += [i_cnd(rm.QM[j], rm.QM[j], rm.Bv[j], S_NEG) for j in range(NL)]
    B.seq += [i_cnd(rm.QT[j], rm.QT[j], rm.A[j], S_NEG) for j in range(NL)]
    for q_, one in ((rm.QP, 1), (rm.QM, 1), (rm.QT, 0)):
        B.seq += [i_cnd(q_[j], one if j == 0 else 0, q_[j], S_NZ) for j in range(NL)]
    return B, FV(rm.QT, max(qt_in.lmax, qn.lmax), max(qt_in.tmax, qn.tmax), max(qt_in.vmax, qn.vmax))


def selftest_member(trials=24, seed=17):
    """the MSM fold's member path on the emulator: (x, y) -> Niels form, sign / past-the-end selection, Niels addition, against the affine law"""
    rng = random.Random(seed)
    rm = RegMap()
    d2v = 2 * D_ED % Q
    l2 = limbs29(d2v)
    S_NZ = "s[44:45]"
    Bm, qp, qt_in = seq_member(rm, FV(S_C, max(l2[:8]), l2[8], d2v))
    Bn, qt = seq_member_select(rm, qt_in, S_NZ)
    Ba = seq_add(rm, True, qt=qt, niels=True, qpm=qp)
    E = Emitter(); E.schedule(Bm.seq); E.schedule(Bn.seq); E.schedule(Ba.seq)
    for t in range(trials):
        P = ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)) if t % 5 else (0, 1)
        z = rng.randrange(1, Q)
        kind = t % 6
        Qp = (0, 1) if kind == 0 else (P if kind == 1 else ((Q - P[0]) % Q, P[1]) if kind == 2 else ED.ed_mul_aff(ED.ED_B, rng.randrange(1, ED.L_ORD)))
        neg, nz = (t // 2) % 2, 0 if t % 7 == 3 else 1
        em = emu_for()
        for j, l in enumerate(l2):
            em.s[S_C[j]] = l
        for regs, v in zip((rm.X1, rm.Y1, rm.Z1, rm.T1), (P[0] * z % Q, P[1] * z % Q, z, P[0] * P[1] * z % Q)):
            em.set9(regs, _rep(rng, v, t % 4 == 3))
        em.set9(rm.A, limbs29(Qp[0] + (Q if t % 3 == 0 else 0))); em.set9(rm.Bv, limbs29(Qp[1]))
        em.c[S_NEG], em.c[S_NZ] = neg, nz
        em.run(E.order)
        gx, gy, gz, gt = (em.get9(r_) % Q for r_ in (rm.X1, rm.Y1, rm.Z1, rm.T1))
        zi = pow(gz, -1, Q)
        member = (0, 1) if not nz else (((Q - Qp[0]) % Q, Qp[1]) if neg else Qp)
        want = ED.ed_add_aff(P, member)
        assert (gx * zi % Q, gy * zi % Q) == want and gt * zi % Q == want[0] * want[1] % Q, ("member", t, kind, neg, nz)
    return True


def emit_msm_acc():
    """Bucket accumulation of the Curve25519 MSM (arkmpc_ed_msm.inc, k_edmsm_accumulate) as one stream: a lane folds the `len` members of its task
    -- vals[first .. first + len): point index | sign << 31, members are 64-byte plain affine (x, y) records -- into an extended sum that starts at
    the identity.  The law is complete: no flags.  A member becomes its Niels form on the fly (y + x, y - x, 2d x y: two products); a lane whose
    task is shorter than the wave's longest adds the identity's Niels form (1, 1, 0) instead.  The next member's index and coordinates are requested
    before the addition.  The sum leaves as the 36-word record of arkmpc_ed_msm.inc (x, y, t, z: nine limbs each, normalised).
    Operands: %[lo4] %[len] (VGPR), %[maxlen] (SGPR), %[vals] %[aff] (SGPR pairs), %[dst] (VGPR pair)."""
    rm = RegMap()
    rec2, last = rm.tid4, rm.tid128
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s_: "%s_%%=" % s_
    S_NZ = "s[44:45]"
    prologue(A)
    d2 = const_to(A, 2 * D_ED % Q)
    for j in range(NL):
        A("v_mov_b32_e32 %s, 0" % rm.X1[j])
        A("v_mov_b32_e32 %s, %d" % (rm.Y1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, %d" % (rm.Z1[j], 1 if j == 0 else 0))
        A("v_mov_b32_e32 %s, 0" % rm.T1[j])

    def load_xy():
        A("v_lshlrev_b32_e32 %s, 6, %s" % (rm.off, rm.rec))                 # 64 bytes per member; the shift drops the sign bit
        ld(rm.ST, rm.off, "aff", 0); ld(rm.LD, rm.off, "aff", 32)

    A("v_add_u32_e32 %s, -1, %%[len]" % last)
    A("global_load_dword %s, %%[lo4], %%[vals]" % rm.rec)
    A("s_waitcnt vmcnt(0)")
    load_xy()
    A("s_mov_b32 %s, 0" % S_STEP)
    EC.align_head(A)
    A(lbl("M_step") + ":")
    A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
    A("v_min_u32_e32 %s, %s, %s" % (rm.tmp, S_TMP, last))
    A("v_lshl_add_u32 %s, %s, 2, %%[lo4]" % (rm.tmp, rm.tmp))
    A("global_load_dword %s, %s, %%[vals]" % (rec2, rm.tmp))
    A("v_cmp_gt_u32_e64 %s, %%[len], %s" % (S_NZ, S_STEP))                  # this lane still has a member at this step
    A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
    A("s_waitcnt vmcnt(1)")
    Bm, qp, qt_in = seq_member(rm, d2, unpack=True)
    _sched(L, Bm.seq)
    # the member after this one: index -> coordinates, in flight during the addition
    A("s_waitcnt vmcnt(0)")
    A("v_mov_b32_e32 %s, %s" % (rm.rec, rec2))
    load_xy()
    Bn, qt = seq_member_select(rm, qt_in, S_NZ)
    _sched(L, Bn.seq, pre={S_NEG: -1, S_NZ: -1})
    Ea = _sched(L, seq_add(rm, True, qt=qt, niels=True, qpm=qp).seq)
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_lt_u32 %s, %%[maxlen]" % S_STEP)
    A("s_cbranch_scc1 " + lbl("M_step"))
    A("s_waitcnt vmcnt(0)")                                                   # the last prefetch (clamped to the last member) is not used
    Bo = Bld(rm)
    for e in acc_fvs(rm):
        Bo.norm(e)
    _sched(L, Bo.seq)
    for k, regs in enumerate((rm.X1, rm.Y1, rm.T1, rm.Z1)):                   # record order x, y, t, z
        for j in range(NL):
            A("global_store_dword %%[dst], %s, off offset:%d" % (regs[j], 4 * (9 * k + j)))
    A("s_waitcnt vmcnt(0)")
    mult = sum(1 for i in Ea.order if i.op in ("mad", "mad24")) + sum(1 for i in Bm.seq if i.op in ("mad", "mad24"))
    return L, rm, dict(add=len(Ea.order), member_mults=mult, vgpr_end=rm.end)


def emit_header(path):
    selftest(trials=14)
    selftest_member(trials=12)
    lines, rm, st = emit_loop()
    out = ["// GENERATED by tools/gen_ed29_asm.py -- do not edit.  The Curve25519 window loop, table kernel and fixed-base chain on NINE 29-bit limbs, plain",
           "// arithmetic mod 2^255 - 19 (2^261 = 1216 folds the high columns of a product into the low ones); operands and memory formats of ed_asm_kernels.inc.",
           "// double: %d instructions (%d with T), add: %d, VGPRs v%d..v%d; %d multiplier instructions per scalar-mul in the loop." %
           (st["double"], st["double_t"], st["add"], rm.first, rm.end - 1, st["loop_mults"]),
           "#pragma once", "#define ED_ASM29_MULT_INSTRS_LOOP %d" % st["loop_mults"],
           "__device__ __forceinline__ void ed_smul_loop29_asm(u32 tid, u32 n, u32 ptid, u32 np, const u64* tab, const u32* dig, u64* res) {", "    asm volatile(",
           G.c_string(lines), "        :", '        : [tid] "v"(tid), [n] "s"(n), [ptid] "v"(ptid), [np] "s"(np), [tab] "s"(tab), [dig] "s"(dig), [res] "s"(res)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(rm.first, rm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    tlines, trm, tst = emit_table()
    out += ["// the window table (cached entries of 0*P .. 16*P): %d asm lines, %d multiplier instructions per scalar-mul" % (len(tlines), tst["table_mults"]),
            "#define ED_ASM29_MULT_INSTRS_TABLE %d" % tst["table_mults"],
            "__device__ __forceinline__ void ed_smul_table29_asm(u32 tid, u32 poff, u32 n, const u64* pts, u64* tab) {", "    asm volatile(",
            G.c_string(tlines), "        :", '        : [tid] "v"(tid), [poff] "v"(poff), [n] "s"(n), [pts] "s"(pts), [tab] "s"(tab)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(trm.first, trm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    glines, grm, gst = emit_gen_chain()
    out += ["// fixed-base multiplication by the base point: %d additions of tabulated affine multiples (plain Niels entries), %d asm lines, %d multiplier instructions" %
            (GEN_WINDOWS, len(glines), gst["chain_mults"]),
            "__device__ __forceinline__ void ed_gen_chain29_asm(u32 tid, u32 n, const u32* dig, const u64* tab, u64* res) {", "    asm volatile(",
            G.c_string(glines), "        :", '        : [tid] "v"(tid), [n] "s"(n), [dig] "s"(dig), [tab] "s"(tab), [res] "s"(res)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(grm.first, grm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    mlines, mrm, mst = emit_msm_acc()
    out += ["// bucket accumulation of the MSM: members become Niels points on the fly, complete addition, no flags: %d asm lines, %d instructions per addition" % (len(mlines), mst["add"]),
            "__device__ __forceinline__ void ed_msm_acc29_asm(u32 lo4, u32 len, u32 maxlen, const u32* vals, const u64* aff, u32* dst) {", "    asm volatile(",
            G.c_string(mlines), "        :", '        : [lo4] "v"(lo4), [len] "v"(len), [maxlen] "s"(maxlen), [vals] "s"(vals), [aff] "s"(aff), [dst] "v"(dst)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mrm.first, mrm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return st, len(lines) + len(tlines) + len(glines) + len(mlines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "ed29_asm_kernels.inc"))
    a = ap.parse_args()
    if a.selftest:
        r = selftest(trials=120)
        selftest_member(trials=60)
        print("multiplications at their operand bounds: %d distinct, ok" % selftest_extremes())
        print("ok:", {k: (len(v[0].order), len(v[1].order)) for k, v in r.items()})
        sys.exit(0)
    st, n = emit_header(a.o)
    print("ed29: %d asm lines; %s" % (n, st))
