#!/usr/bin/env python3
"""Generator for the 29-bit-limb form of the BN254 G1 scalar-multiplication kernels (ark-mpc_amd/csrc/ec29_asm_kernels.inc).

Same algorithm as tools/gen_ec_asm.py (effective-affine window table, blinded accumulator, dbl-2009-l / madd-2007-bl, GLV halves, one
per-lane exceptional flag), on a different number representation: a field element is NINE unsaturated limbs of 29 bits in nine VGPRs and
the Montgomery radix is R' = 2^261.

  * The multiplier is product scanning: every column of sum a_i b_j + sum m_i q_j accumulates in ONE 64-bit VGPR pair through the 64-bit
    addend of v_mad_u64_u32; nine 58-bit products, nine reduction products and a carry stay below 2^64, so there is no carry handling
    at all -- a column ends with one 64-bit shift.  171 multiplier + 44 other instructions, against 136 + 162 for the 32-bit CIOS block:
    872 against 1184 cycles per wave-multiplication on MI355X (probes/mulrate29.hip, profiles/r03_ec/mulrate29.jsonl).
  * Additions, subtractions, doublings are limb-wise, carry-free: a - b is a + K - b with K a multiple of q whose limbs dominate b's
    (value grows by K, limbs by K's).  Values and limbs are only bounded, not reduced; `norm` is one parallel carry pass (limbs back
    below 2^29 + a few), and every multiplication brings the VALUE back below 2q.  The generator carries exact bounds (largest limb, largest
    top limb, largest value) through both bodies and asserts that every multiplier column fits 64 bits, every limb 32 bits, and that the
    accumulator invariant reproduces itself -- the bodies cannot overflow for ANY input.
  * No zero tests in the loop: if an addition hits H = 0 (mod q) its Z3 = 2 Z1 H is 0 (mod q), and Z stays 0 (mod q) through every later
    doubling / addition, so the exceptional flag is one test of the final Z in the epilogue (same lanes as the 32-bit loop flags: the
    finish kernel recomputes them on the compiled path).
  * Memory formats do not change: table entries and the loop's result are packed into 8 x 32-bit words (values below 2q < 2^255), the
    result and the common Z are converted back to R = 2^256 by one multiplication with a constant, so k_g1_smul_finish and the digit kernel
    are shared with the 32-bit pipeline.  Only the table kernel's private Jacobian scratch (jtab) holds 27-word entries (112-byte stride).

Both bodies are executed by the single-lane emulator against the affine group law in Python integers before they are emitted, with
random and with extreme representations of the inputs (tests/test_asm_generator.py).

Reference semantics: CurvePoint * Scalar (online-phase/src/algebra/curve/curve.rs:403-409), PointShare * Scalar (curve/share.rs:108-114).
"""
import argparse
import hashlib
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_asm_kernels as G
import gen_ec_asm as EC
from gen_asm_kernels import Ins, Emitter, Emu, M32, regs_of, pr, src

Q, RORD = EC.Q, EC.RORD
NL = 9
M29 = (1 << 29) - 1
RP = 1 << 261                       # Montgomery radix of this representation
R32 = 1 << 256                      # radix of the engine's 32-bit representation (memory formats)
N_STEPS, N_TABLE = EC.N_STEPS, EC.N_TABLE
JT_STRIDE = 128                     # bytes per scratch entry of the table kernel: eight 16-byte chunks (27 words used)

# ---- SGPR map (all clobbered by the asm statements) -------------------------------------------------------------------------
S_JUNK, S_INV, S_MASK = "s[16:17]", "s18", "s19"
S_Q = ["s%d" % (20 + i) for i in range(NL)]           # q limbs (29-bit)
S_C = ["s%d" % (29 + i) for i in range(NL)]           # a constant operand: beta, conversion constants, R0 coordinates
S_STEP, S_N4, S_N64, S_TMP, S_DBL, S_NJT, S_IDX = "s38", "s39", "s40", "s41", "s42", "s43", "s56"
S_NZ, S_NEG, S_EXC, S_M1, S_M2 = "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "s[52:53]"
CLOBBER_SGPRS = ["s%d" % i for i in range(16, 58)]


def limbs29(v):
    """canonical limbs of a value below 2^(232 + 32): eight 29-bit limbs and the rest"""
    return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]


def val29(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


mont = lambda v: v * RP % Q
unmont = lambda v: v * pow(RP, -1, Q) % Q


# ---- instruction constructors ------------------------------------------------------------------------------------------------
def i_mad(d, a, b, c):
    return Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(d), S_JUNK, a, b, "0" if c == 0 else pr(c)), "mad", (d, a, b, c),
               rd=regs_of(a, b, c if c != 0 else None), wr=list(d))
def i_mul_lo(d, a, b): return Ins("v_mul_lo_u32 %s, %s, %s" % (d, a, b), "mul_lo", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_and(d, a, b): return Ins("v_and_b32_e32 %s, %s, %s" % (d, src(a), b), "and", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_shr64(d, s_, sh): return Ins("v_lshrrev_b64 %s, %d, %s" % (pr(d), sh, pr(s_)), "shr64", (d, s_, sh), rd=list(s_), wr=list(d))
def i_mov(d, s_): return Ins("v_mov_b32_e32 %s, %s" % (d, src(s_)), "mov", (d, s_), rd=regs_of(s_), wr=[d])
def i_add(d, a, b): return Ins("v_add_u32_e32 %s, %s, %s" % (d, src(a), b), "add", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_sub(d, a, b): return Ins("v_sub_u32_e32 %s, %s, %s" % (d, src(a), b), "sub", (d, a, b), rd=regs_of(a, b), wr=[d])          # a - b
def i_shl(d, a, sh): return Ins("v_lshlrev_b32_e32 %s, %d, %s" % (d, sh, a), "shl", (d, a, sh), rd=regs_of(a), wr=[d])
def i_lshr(d, a, sh): return Ins("v_lshrrev_b32_e32 %s, %d, %s" % (d, sh, a), "lshr", (d, a, sh), rd=regs_of(a), wr=[d])
def i_lshl_add(d, a, sh, c): return Ins("v_lshl_add_u32 %s, %s, %d, %s" % (d, a, sh, c), "lshl_add", (d, a, sh, c), rd=regs_of(a, c), wr=[d])
def i_lshl_or(d, a, sh, c): return Ins("v_lshl_or_b32 %s, %s, %d, %s" % (d, a, sh, c), "lshl_or", (d, a, sh, c), rd=regs_of(a, c), wr=[d])
def i_alignbit(d, hi, lo, sh): return Ins("v_alignbit_b32 %s, %s, %s, %d" % (d, hi, lo, sh), "alignbit", (d, hi, lo, sh), rd=regs_of(hi, lo), wr=[d])
def i_or(d, a, b): return Ins("v_or_b32_e32 %s, %s, %s" % (d, src(a), b), "or", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_xor(d, a, b): return Ins("v_xor_b32_e32 %s, %s, %s" % (d, src(a), b), "xor", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_cmpz(mask, a): return Ins("v_cmp_eq_u32_e64 %s, 0, %s" % (mask, a), "cmpz", (mask, a), swr=(mask,), rd=regs_of(a))
def i_cnd(d, f, t, cy): return Ins("v_cndmask_b32_e64 %s, %s, %s, %s" % (d, f, t, cy), "cnd", (d, f, t, cy), srd=(cy,), rd=regs_of(f, t), wr=[d])   # cy ? t : f


class Emu29(Emu):
    """single-lane emulator; multiplier columns and limb additions are checked for overflow as they execute"""

    def run(self, order):
        for ins in order:
            op, a = ins.op, ins.args
            if op == "mad":
                d, x, y, c = a
                cv = 0 if c == 0 else (self.rd(c[0]) | (self.rd(c[1]) << 32))
                r = self.rd(x) * self.rd(y) + cv
                assert r < (1 << 64), "multiplier column overflows 64 bits"
                self.v[d[0]], self.v[d[1]] = r & M32, r >> 32
            elif op == "shr64":
                d, s_, sh = a
                v = (self.rd(s_[0]) | (self.rd(s_[1]) << 32)) >> sh
                self.v[d[0]], self.v[d[1]] = v & M32, v >> 32
            elif op == "add":
                self.v[a[0]] = (self.rd(a[1]) + self.rd(a[2])) & M32
            elif op == "sub":
                self.v[a[0]] = (self.rd(a[1]) - self.rd(a[2])) & M32
            elif op == "shl":
                r = self.rd(a[1]) << a[2]
                assert r <= M32, "limb shift overflows"
                self.v[a[0]] = r
            elif op == "lshr":
                self.v[a[0]] = self.rd(a[1]) >> a[2]
            elif op == "lshl_add":
                r = (self.rd(a[1]) << a[2]) + self.rd(a[3])
                assert r <= M32, "limb shift-add overflows"
                self.v[a[0]] = r
            elif op == "lshl_or":
                self.v[a[0]] = ((self.rd(a[1]) << a[2]) & M32) | self.rd(a[3])
            elif op == "alignbit":
                self.v[a[0]] = (((self.rd(a[1]) << 32) | self.rd(a[2])) >> a[3]) & M32
            elif op == "or":
                self.v[a[0]] = self.rd(a[1]) | self.rd(a[2])
            elif op == "xor":
                self.v[a[0]] = self.rd(a[1]) ^ self.rd(a[2])
            elif op == "cmpz":
                self.c[a[0]] = 1 if self.rd(a[1]) == 0 else 0
            else:
                Emu.run(self, [ins])

    def set9(self, regs, limbs):
        for r, l in zip(regs, limbs):
            self.v[r] = l

    def get9(self, regs):
        return val29([self.v[r] for r in regs])


# ---- values with bounds ------------------------------------------------------------------------------------------------------
class FV:
    """a field element in nine registers: limbs 0..7 <= lmax, limb 8 <= tmax, value <= vmax"""
    __slots__ = ("r", "lmax", "tmax", "vmax")

    def __init__(self, r, lmax, tmax, vmax):
        self.r, self.lmax, self.vmax = list(r), lmax, vmax
        self.tmax = min(tmax, vmax >> 232)                 # limbs are non-negative: the top limb cannot exceed value >> 232
        assert self.lmax <= M32 and self.tmax <= M32

    @property
    def big(self):
        return max(self.lmax, self.tmax)


def fv_mulout(r, vmax_prod):
    v = (vmax_prod >> 261) + Q                              # REDC(x) < x / R' + q
    return FV(r, M29, v >> 232, v)


def fv_const(sregs, value):
    l = limbs29(value)
    return FV(sregs, max(l[:8]), l[8], value)


def fv_acc(r):
    """the invariant of the accumulator's X and Y between bodies"""
    return FV(r, ACC_LMAX, ACC_VMAX >> 232, ACC_VMAX)


ACC_LMAX = M29 + 8
ACC_VMAX = 8 * Q
_K_cache = {}


def k_for(b):
    """limbs of a multiple of q that dominate every limb of b: k_i >= b.lmax (i < 8), k_8 >= b.tmax"""
    u = -(-b.lmax // M29)                                   # borrow u from each higher limb: k_i = c_i - u + u 2^29 >= u (2^29 - 1)
    key = (u, b.tmax)
    if key not in _K_cache:
        c = max(1, ((b.tmax + u) << 232) // Q)
        while True:
            l = limbs29(c * Q)
            if l[8] - u >= b.tmax:
                break
            c += 1
        k = [l[0] + (u << 29)] + [l[i] - u + (u << 29) for i in range(1, 8)] + [l[8] - u]
        assert val29(k) == c * Q and min(k) >= 0 and all(k[i] >= b.lmax for i in range(8))
        _K_cache[key] = k
    return _K_cache[key]


class Bld:
    """builds an instruction sequence while carrying the bounds"""

    def __init__(self, rm):
        self.rm, self.seq = rm, []
        self.n_mul = self.n_sqr = 0
        self.sigs = []                                      # operand bounds of every multiplication built (selftest_extremes)

    # -- multiplier ---------------------------------------------------------------------------------------------------------
    def mul(self, prods, out):
        """out = sum a_k b_k / R' (mod q), value < sum a b / R' + q; out may alias any operand.  A pair (a, a) is a squaring."""
        rm = self.rm
        acc, m, W = rm.acc, rm.m, rm.W
        col = 9 * M29 * M29 + (1 << 36)
        vprod = 0
        plist = []                                          # (x regs, y regs, squaring?)
        nsq = 0
        for a, b in prods:
            if a is b:
                assert 2 * a.big <= M32
                nsq += 1
                assert nsq == 1, "one doubled operand register set"
                self.seq += [i_shl(W[j], a.r[j], 1) for j in range(NL)]
                plist.append((a.r, W, True))
            else:
                plist.append((a.r, b.r, False))
            col += 9 * a.big * b.big
            vprod += a.vmax * b.vmax
        self.sigs.append(tuple((a.lmax, a.tmax, b.lmax, b.tmax, a is b) for a, b in prods))
        assert col < (1 << 64), "multiplier column can overflow: normalise an operand (%.2f bits)" % (col.bit_length())
        first = True
        for k in range(2 * NL - 1):
            terms = []
            for x, y, sq in plist:
                for i in range(NL):
                    j = k - i
                    if not 0 <= j < NL:
                        continue
                    if sq:
                        if i < j: terms.append((x[i], y[j]))          # a_i * (2 a_j)
                        elif i == j: terms.append((x[i], x[i]))
                    else:
                        terms.append((x[i], y[j]))
            terms += [(m[i], S_Q[k - i]) for i in range(NL) if 0 <= k - i < NL and i < k]     # m_k q_0 joins after m_k exists
            for x, y in terms:
                self.seq.append(i_mad(acc, x, y, 0 if first else acc))
                first = False
            if k < NL:
                self.seq += [i_mul_lo(m[k], acc[0], S_INV), i_and(m[k], S_MASK, m[k]), i_mad(acc, m[k], S_Q[0], acc), i_shr64(acc, acc, 29)]
            else:
                self.seq += [i_and(out[k - NL], S_MASK, acc[0]), i_shr64(acc, acc, 29)]
        self.seq.append(i_mov(out[NL - 1], acc[0]))
        if nsq: self.n_sqr += 1
        else: self.n_mul += 1
        return fv_mulout(out, vprod)

    # -- carry-free limb operations ---------------------------------------------------------------------------------------------
    def add(self, a, b, out):
        self.seq += [i_add(out[j], a.r[j], b.r[j]) for j in range(NL)]
        return FV(out, a.lmax + b.lmax, a.tmax + b.tmax, a.vmax + b.vmax)

    def shl(self, a, sh, out):
        self.seq += [i_shl(out[j], a.r[j], sh) for j in range(NL)]
        return FV(out, a.lmax << sh, a.tmax << sh, a.vmax << sh)

    def shl_add(self, a, sh, c, out):                                  # (a << sh) + c
        self.seq += [i_lshl_add(out[j], a.r[j], sh, c.r[j]) for j in range(NL)]
        return FV(out, (a.lmax << sh) + c.lmax, (a.tmax << sh) + c.tmax, (a.vmax << sh) + c.vmax)

    def sub(self, a, b, out):
        """a - b + K(b); out may alias a or b"""
        k = k_for(b)
        for j in range(NL):
            self.seq += [i_sub(out[j], a.r[j], b.r[j]), i_add(out[j], k[j], out[j])]
        return FV(out, a.lmax + max(k[:8]), a.tmax + k[8], a.vmax + val29(k))

    def neg(self, b, out):
        k = k_for(b)
        self.seq += [i_sub(out[j], k[j], b.r[j]) for j in range(NL)]
        return FV(out, max(k[:8]), k[8], val29(k))

    def norm(self, a, out=None):
        """one parallel carry pass: limbs <= 2^29 - 1 + (lmax >> 29); in place unless out is given"""
        out = out or a.r
        W = self.rm.W
        self.seq += [i_lshr(W[j], a.r[j], 29) for j in range(8)]
        self.seq += [i_and(out[0], S_MASK, a.r[0])]
        for j in range(1, 8):
            self.seq += [i_and(out[j], S_MASK, a.r[j]), i_add(out[j], W[j - 1], out[j])]
        self.seq += [i_add(out[8], W[7], a.r[8])]
        c = a.lmax >> 29
        return FV(out, M29 + c, a.tmax + c, a.vmax)

    def movs(self, a, out):
        self.seq += [i_mov(out[j], a.r[j]) for j in range(NL)]
        return FV(out, a.lmax, a.tmax, a.vmax)


def check_acc(v, what):
    assert v.lmax <= ACC_LMAX and v.vmax <= ACC_VMAX, "%s leaves the accumulator invariant: limbs %d (+%d), value %.2f q" % (
        what, v.lmax, v.lmax - M29, v.vmax / Q)


def fv_z(r):
    return fv_mulout(r, (20 * Q) * (2 * Q))                 # Z is always a multiplier output; 20q x 2q covers every producer below


# ---- the two bodies ----------------------------------------------------------------------------------------------------------
def seq_double(rm, B=None):
    """dbl-2009-l with a = 0, in place on (X1, Y1, Z1); scratch X2, Y2, SX, T0..T2, W.  D = X (4B) keeps the powers of two in limb
    shifts and Y3 = E (D - X3) + (K - 4B)(2B) is ONE reduction of two products; 6 multiplier blocks (3 squarings), 3 subtractions / negations,
    4 carry passes."""
    B = B or Bld(rm)
    X, Y, Z = fv_acc(rm.X1), fv_acc(rm.Y1), fv_z(rm.Z1)
    A = B.mul([(X, X)], rm.T0)
    Bq = B.mul([(Y, Y)], rm.T1)
    Y2 = B.shl(Y, 1, rm.Y1)
    Z3 = B.mul([(Y2, Z)], rm.Z1)
    B4 = B.shl(Bq, 2, rm.T2)
    D = B.mul([(X, B4)], rm.X2)
    E = B.norm(B.shl_add(A, 1, A, rm.SX))                   # E = 3 A
    NB4 = B.norm(B.neg(B4, rm.T2))                          # K - 4 B
    B2 = B.shl(Bq, 1, rm.T1)
    F = B.mul([(E, E)], rm.Y2)
    D2 = B.shl(D, 1, rm.T0)
    X3 = B.norm(B.sub(F, D2, rm.X1))
    DX = B.norm(B.sub(D, X3, rm.X2))
    Y3 = B.mul([(E, DX), (NB4, B2)], rm.Y1)                 # E (D - X3) - 8 B^2: one reduction for both products
    check_acc(X3, "double X3"); check_acc(Y3, "double Y3")
    assert Z3.vmax <= fv_z(rm.Z1).vmax
    return B


def seq_madd(rm, B=None, y2=None, h2_out=None):
    """madd-2007-bl: (X1, Y1, Z1) += affine (X2, Y2), in place; scratch T0..T2, W.  Z3 = Z1 (2H), I = (2H)^2, and
    Y3 = r (V - X3) + (K - 2 Y1) J is ONE reduction of two products.  10 multiplier blocks (3 squarings), 5 subtractions / negations,
    5 carry passes.  y2: bounds of the second operand's y (the loop negates it conditionally).  h2_out: registers that keep 2H = Z3 / Z1
    (the table kernel rescales its entries with these factors)."""
    B = B or Bld(rm)
    B.h2 = None
    X1, Y1, Z1 = fv_acc(rm.X1), fv_acc(rm.Y1), fv_z(rm.Z1)
    X2 = fv_mulout(rm.X2, (2 * Q) * (2 * Q))                # table entries are multiplier outputs (below 2q, limbs below 2^29)
    Y2 = y2 or fv_mulout(rm.Y2, (2 * Q) * (2 * Q))
    ZZ = B.mul([(Z1, Z1)], rm.T0)
    U2 = B.mul([(X2, ZZ)], rm.X2)
    YZ = B.mul([(Y2, Z1)], rm.Y2)
    S2 = B.mul([(YZ, ZZ)], rm.Y2)
    H = B.norm(B.sub(U2, X1, rm.X2))
    H2 = B.shl(H, 1, h2_out or rm.T0)
    Z3 = B.mul([(Z1, H2)], rm.Z1)
    B.h2 = H2
    I = B.mul([(H2, H2)], rm.T0)
    r = B.shl(B.norm(B.sub(S2, Y1, rm.Y2)), 1, rm.Y2)
    V = B.mul([(X1, I)], rm.X1)
    J = B.mul([(H, I)], rm.X2)
    RR = B.mul([(r, r)], rm.T0)
    JV = B.shl_add(V, 1, J, rm.T1)                          # J + 2 V
    X3 = B.norm(B.sub(RR, JV, rm.T0))
    VX = B.norm(B.sub(V, X3, rm.X1))
    NY = B.norm(B.neg(B.shl(Y1, 1, rm.Y1), rm.Y1))          # K - 2 Y1
    Y3 = B.mul([(r, VX), (NY, J)], rm.Y1)
    X3 = B.movs(X3, rm.X1)
    check_acc(X3, "madd X3"); check_acc(Y3, "madd Y3")
    assert Z3.vmax <= fv_z(rm.Z1).vmax
    return B


# ---- packing: a value below 2^256 in normalised limbs <-> 8 x 32-bit words -------------------------------------------------------
def seq_unpack(words, out):
    """8 words -> 9 limbs (out must not overlap words)"""
    s = [i_and(out[0], S_MASK, words[0])]
    for i in range(1, 8):
        bit = 29 * i
        w, sh = bit >> 5, bit & 31
        s += [i_alignbit(out[i], words[w + 1], words[w], sh), i_and(out[i], S_MASK, out[i])]
    s += [i_lshr(out[8], words[7], 8)]
    return s


def seq_pack(l, words):
    """9 normalised limbs (limbs < 2^29, value < 2^256) -> 8 words (words must not overlap l)"""
    s = []
    for w in range(8):
        lo = 32 * w
        first = True
        for i in range(NL):
            b0 = 29 * i
            if b0 + 29 <= lo or b0 >= lo + 32:
                continue
            if b0 <= lo:
                s.append(i_lshr(words[w], l[i], lo - b0))
            else:
                assert not first
                s.append(i_lshl_or(words[w], l[i], b0 - lo, words[w]))
            first = False
    return s


# ---- registers -------------------------------------------------------------------------------------------------------------------
class RegMap:
    """Fixed VGPR allocation (v8 upwards).  All three kernels run at three waves per SIMD; the streams are VALU-issue-bound from two waves up
    (PMC, config 4: 4.08 SIMD cycles per VALU instruction in the loop AND in the table kernel -- one wave-instruction per 4 cycles is the
    hardware's limit; a table kernel squeezed into 120 VGPRs for four waves per SIMD measured no faster)."""

    def __init__(self, first=8, table_kernel=False, msm=False):
        rg = G.Regs(first)
        blk = rg.vec(27, 2)
        self.X1, self.Y1, self.Z1 = blk[0:9], blk[9:18], blk[18:27]          # accumulator; contiguous: one jtab entry
        blk = rg.vec(27, 2)
        self.X2, self.Y2, self.SX = blk[0:9], blk[9:18], blk[18:27]          # table entry / operand; contiguous for the table kernel
        blk = rg.vec(18, 2)
        self.SY, self.SZ = blk[0:9], blk[9:18]                                # loop: saved accumulator; table kernel: (x, y) of P, second entry buffer
        self.T0, self.T1, self.T2 = rg.vec(9, 2), rg.vec(9, 2), rg.vec(9, 2)
        self.W = rg.vec(9, 2)                                                # doubled operand of a squaring / carries of norm
        self.m = rg.vec(9)
        self.acc = rg.pair()
        self.off, self.tid64, self.tmp = (rg.one() for _ in range(3))
        if table_kernel:
            self.off2, self.tid32, self.tidjt = (rg.one() for _ in range(3))
        else:
            self.rec, self.tid4, self.tid96, self.flag = (rg.one() for _ in range(4))
        if msm:
            self.LD1, self.LD2 = rg.vec(8, 2), rg.vec(8, 2)                  # landing registers of the prefetched member (the addition uses T1 / T2)
            self.rec2, self.last = rg.one(), rg.one()
        self.first, self.end = first, rg.next


def vrange(regs):
    a = int(regs[0][1:])
    assert a % 2 == 0 and [int(r[1:]) for r in regs] == list(range(a, a + len(regs))), regs
    return "v[%d:%d]" % (a, a + len(regs) - 1) if len(regs) > 1 else regs[0]


def mem_ops(A):
    """load / store helpers over a line sink A: a run of consecutive registers as dwordx4 / x3 / x2 / x1 pieces"""
    def pieces(regs, byte):
        out, i = [], 0
        while i < len(regs):
            b = byte + 4 * i
            n = min(4, len(regs) - i) if b % 16 == 0 else (min(2, len(regs) - i) if b % 8 == 0 else 1)
            out.append((regs[i:i + n], b))
            i += n
        return out
    sfx = {1: "dword", 2: "dwordx2", 3: "dwordx3", 4: "dwordx4"}

    def ld(regs, off, base, byte=0):
        for rs, b in pieces(regs, byte):
            A("global_load_%s %s, %s, %%[%s] offset:%d" % (sfx[len(rs)], vrange(rs), off, base, b))

    def st(regs, off, base, byte=0):
        for rs, b in pieces(regs, byte):
            A("global_store_%s %s, %s, %%[%s] offset:%d" % (sfx[len(rs)], off, vrange(rs), base, b))
        A("s_nop 0")                                       # a wide store's data registers may not be rewritten by the very next VALU instruction
    return ld, st


def prologue(A, rm):
    inv = (-pow(Q, -1, 1 << 29)) % (1 << 29)
    A("s_nop 1")
    A("s_mov_b32 %s, 0x%08x" % (S_INV, inv))
    A("s_mov_b32 %s, 0x%08x" % (S_MASK, M29))
    for j, l in enumerate(limbs29(Q)):
        A("s_mov_b32 %s, 0x%08x" % (S_Q[j], l))


def const_to(A, value):
    for j, l in enumerate(limbs29(value)):
        A("s_mov_b32 %s, 0x%08x" % (S_C[j], l))
    return fv_const(S_C, value)


def emu_for():
    em = Emu29()
    em.s[S_INV] = (-pow(Q, -1, 1 << 29)) % (1 << 29)
    em.s[S_MASK] = M29
    for j, l in enumerate(limbs29(Q)):
        em.s[S_Q[j]] = l
    return em


# ---- self-test ---------------------------------------------------------------------------------------------------------------------
def build_body(which, sched=True):
    rm = RegMap()
    B = seq_double(rm) if which == "double" else seq_madd(rm)
    E = Emitter()
    (E.schedule if sched else E.emit_all)(B.seq)
    return E, rm, B


def _rep(rng, v, vmax, lmax, extreme=False):
    """a representation of v (mod q): value below vmax, limbs below lmax (>= 2^29 - 1)"""
    cmax = (vmax - v) // Q
    val = v + (cmax if extreme else rng.randrange(cmax + 1)) * Q
    l = limbs29(val)
    slack = lmax - M29
    # a limb above 2^29 - 1 arises as (limb + 2^29) with the next limb one lower; possible where the limb is at most `slack`
    for i in range(8):
        if l[i] <= slack and l[i + 1] > 0 and (extreme or rng.random() < 0.5):
            l[i] += 1 << 29
            l[i + 1] -= 1
    assert val29(l) == val and max(l[:8]) <= lmax
    return l


def _affine(X, Y, Z):
    X, Y, Z = unmont(X % Q), unmont(Y % Q), unmont(Z % Q)
    if Z == 0: return None
    zi = pow(Z, -1, Q)
    return (X * zi * zi % Q, Y * zi * zi * zi % Q)


def selftest(trials=40, seed=11):
    rng = random.Random(seed)
    Ed, rm, _ = build_body("double")
    Ea, rm2, _ = build_body("madd")
    zmax = fv_z(rm.Z1).vmax
    for t in range(trials):
        ext = t % 5 == 4
        P = EC.g1_mul(EC.GEN, rng.randrange(1, RORD))
        z = rng.randrange(1, Q)
        Xj, Yj = P[0] * z * z % Q, P[1] * z * z * z % Q
        em = emu_for()
        em.set9(rm.X1, _rep(rng, mont(Xj), ACC_VMAX, ACC_LMAX, ext)); em.set9(rm.Y1, _rep(rng, mont(Yj), ACC_VMAX, ACC_LMAX, ext))
        em.set9(rm.Z1, _rep(rng, mont(z), zmax, M29, ext))
        em.run(Ed.order)
        got = [[em.v[r] for r in regs] for regs in (rm.X1, rm.Y1, rm.Z1)]
        assert all(max(l[:8]) <= ACC_LMAX for l in got) and max(val29(got[0]), val29(got[1])) <= ACC_VMAX and val29(got[2]) <= zmax, "double: bounds"
        assert _affine(*[val29(l) for l in got]) == EC.g1_add(P, P), ("double", t)
        kind = t % 8
        Qp = P if kind == 6 else ((P[0], Q - P[1]) if kind == 7 else EC.g1_mul(EC.GEN, rng.randrange(1, RORD)))
        em = emu_for()
        em.set9(rm2.X1, _rep(rng, mont(Xj), ACC_VMAX, ACC_LMAX, ext)); em.set9(rm2.Y1, _rep(rng, mont(Yj), ACC_VMAX, ACC_LMAX, ext))
        em.set9(rm2.Z1, _rep(rng, mont(z), zmax, M29, ext))
        em.set9(rm2.X2, _rep(rng, mont(Qp[0]), 2 * Q, M29, ext)); em.set9(rm2.Y2, _rep(rng, mont(Qp[1]), 2 * Q, M29, ext))
        em.run(Ea.order)
        got = [[em.v[r] for r in regs] for regs in (rm2.X1, rm2.Y1, rm2.Z1)]
        assert all(max(l[:8]) <= ACC_LMAX for l in got) and max(val29(got[0]), val29(got[1])) <= ACC_VMAX and val29(got[2]) <= zmax, "madd: bounds"
        if kind >= 6:
            assert val29(got[2]) % Q == 0, ("madd: H = 0 must leave Z = 0 (mod q)", t)
            # ... and Z stays 0 (mod q) through a following doubling and addition (what the epilogue's single test relies on)
            em2 = emu_for()
            for regs, l in zip((rm.X1, rm.Y1, rm.Z1), got): em2.set9(regs, l)
            em2.run(Ed.order)
            assert em2.get9(rm.Z1) % Q == 0
            em3 = emu_for()
            for regs, l in zip((rm2.X1, rm2.Y1, rm2.Z1), got): em3.set9(regs, l)
            em3.set9(rm2.X2, limbs29(mont(Qp[0]))); em3.set9(rm2.Y2, limbs29(mont(Qp[1])))
            em3.run(Ea.order)
            assert em3.get9(rm2.Z1) % Q == 0
        else:
            assert _affine(*[val29(l) for l in got]) == EC.g1_add(P, Qp), ("madd", t)
    # packing round trip
    rmx = RegMap()
    for t in range(50):
        v = rng.randrange(2 * Q) if t else 2 * Q - 1
        em = emu_for()
        em.set9(rmx.X1, limbs29(v))
        E = Emitter(); E.schedule(seq_pack(rmx.X1, rmx.T1[:8]) + seq_unpack(rmx.T1[:8], rmx.X2))
        em.run(E.order)
        assert sum(em.v[r] << (32 * i) for i, r in enumerate(rmx.T1[:8])) == v and em.get9(rmx.X2) == v
    return Ed, Ea


def selftest_extremes():
    """Every multiplication the two bodies contain, re-built on its own with ALL limbs of both operands at the bounds the generator
    carried to that point: the emulator asserts that no column overflows 64 bits and the result is checked against Python integers."""
    rm = RegMap()
    sigs = set(seq_double(rm).sigs) | set(seq_madd(rm).sigs)
    Bn = Bld(rm)
    yn = Bn.neg(fv_mulout(rm.Y2, 4 * Q * Q), rm.T2)
    sigs |= set(seq_madd(rm, y2=FV(rm.Y2, yn.lmax, yn.tmax, yn.vmax)).sigs)
    Rinv = pow(RP, -1, Q)
    for sig in sorted(sigs):
        B = Bld(rm)
        em = emu_for()
        ops, want = [], 0
        pool = [rm.X1, rm.X2, rm.Y2, rm.SX]
        for k, (al, at, bl, bt, sq) in enumerate(sig):
            ra, rb = pool[2 * k], pool[2 * k + 1]
            la, lb = [al] * 8 + [at], [bl] * 8 + [bt]
            a = FV(ra, al, at, val29(la)); em.set9(ra, la)
            if sq:
                ops.append((a, a)); want += val29(la) ** 2
            else:
                b = FV(rb, bl, bt, val29(lb)); em.set9(rb, lb)
                ops.append((a, b)); want += val29(la) * val29(lb)
        B.mul(ops, rm.Y1)
        E = Emitter(); E.schedule(B.seq)
        em.run(E.order)
        got = [em.v[r] for r in rm.Y1]
        assert max(got[:8]) <= M29 and val29(got) % Q == want * Rinv % Q and val29(got) <= want // RP + Q, sig
    return len(sigs)


# ---- the window loop -----------------------------------------------------------------------------------------------------------------
def emit_loop():
    """Operands as g1_smul_loop_asm (tools/gen_ec_asm.py): %[tid] %[ptid] (VGPR), %[n] %[np] (SGPR), %[tab] %[dig] %[res] %[exc] (SGPR pairs).
    The digit record and the table entry of step s + 1 are requested before the addition of step s (landing registers LD1 / LD2, which no
    body touches), so a step never waits for memory."""
    rm = RegMap(msm=True)
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s: "%s_%%=" % s

    def sched(seq, pre=None):
        E = Emitter()
        if pre: E.lastw.update(pre)
        E.schedule(seq)
        L.extend(E.lines)
        return E

    def load_entry():                                                           # table entry named by rm.rec -> LD1 / LD2
        A("v_and_b32_e32 %s, 31, %s" % (rm.tmp, rm.rec))
        A("v_mul_lo_u32 %s, %s, %s" % (rm.tmp, rm.tmp, S_N64))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, rm.tmp, rm.tid64))
        ld(rm.LD1, rm.off, "tab", 0); ld(rm.LD2, rm.off, "tab", 32)

    prologue(A, rm)
    beta = const_to(A, mont(EC.unmont(EC.BETA)))           # stays in S_C until the epilogue's constant replaces it
    A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
    A("v_lshlrev_b32_e32 %s, 6, %%[ptid]" % rm.tid64)
    A("v_mul_u32_u24_e32 %s, 96, %%[tid]" % rm.tid96)
    A("s_lshl_b32 %s, %%[n], 2" % S_N4)
    A("s_lshl_b32 %s, %%[np], 6" % S_N64)
    A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.tid4))                   # record of step 0
    # accumulator = table entry 16 (R0 on the isomorphic curve), Z = 1 (Montgomery form)
    A("s_mul_i32 %s, %s, 16" % (S_TMP, S_N64))
    A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid64))
    ld(rm.T1[:8], rm.off, "tab", 0); ld(rm.T2[:8], rm.off, "tab", 32)
    for j, l in enumerate(limbs29(RP % Q)):
        A("v_mov_b32_e32 %s, 0x%08x" % (rm.Z1[j], l))
    A("s_waitcnt vmcnt(0)")
    load_entry()
    sched(seq_unpack(rm.T1[:8], rm.X1) + seq_unpack(rm.T2[:8], rm.Y1))
    A("s_mov_b32 %s, 0" % S_STEP)
    EC.align_head(A)
    A(lbl("L_step") + ":")
    # record of the NEXT step (the last step asks for its own again)
    A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
    A("s_min_u32 %s, %s, %d" % (S_TMP, S_TMP, N_STEPS - 1))
    A("s_mul_i32 %s, %s, %s" % (S_TMP, S_TMP, S_N4))
    A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid4))
    A("global_load_dword %s, %s, %%[dig]" % (rm.rec2, rm.off))
    # five doublings before the first half of every window but the top one (steps 2, 4, ..., 52)
    A("s_and_b32 %s, %s, 1" % (S_TMP, S_STEP))
    A("s_cmp_eq_u32 %s, 1" % S_TMP)
    A("s_cbranch_scc1 " + lbl("L_nodbl"))
    A("s_cmp_eq_u32 %s, 0" % S_STEP)
    A("s_cbranch_scc1 " + lbl("L_nodbl"))
    A("s_cmp_eq_u32 %s, %d" % (S_STEP, N_STEPS - 1))
    A("s_cbranch_scc1 " + lbl("L_nodbl"))
    A("s_mov_b32 %s, 5" % S_DBL)
    EC.align_head(A)
    A(lbl("L_dbl") + ":")
    Bd = seq_double(rm)
    Ed = sched(Bd.seq)
    A("s_sub_u32 %s, %s, 1" % (S_DBL, S_DBL))
    A("s_cmp_lg_u32 %s, 0" % S_DBL)
    A("s_cbranch_scc1 " + lbl("L_dbl"))
    A(lbl("L_nodbl") + ":")
    # masks from this step's record (bits 0-4 table index, bit 5 negate, bit 6 digit non-zero); the accumulator is saved
    A("v_and_b32_e32 %s, 32, %s" % (rm.tmp, rm.rec))
    A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NEG, rm.tmp))
    A("v_and_b32_e32 %s, 64, %s" % (rm.tmp, rm.rec))
    A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NZ, rm.tmp))
    for d, s_ in zip(rm.SX + rm.SY + rm.SZ, rm.X1 + rm.Y1 + rm.Z1):
        A("v_mov_b32_e32 %s, %s" % (d, s_))
    A("s_waitcnt vmcnt(0)")
    sched(seq_unpack(rm.LD1, rm.X2) + seq_unpack(rm.LD2, rm.Y2))
    # the next step's entry flies during this step's addition
    A("v_mov_b32_e32 %s, %s" % (rm.rec, rm.rec2))
    load_entry()
    # phi half (odd steps): x -> beta x
    A("s_and_b32 %s, %s, 1" % (S_TMP, S_STEP))
    A("s_cmp_eq_u32 %s, 0" % S_TMP)
    A("s_cbranch_scc1 " + lbl("L_nobeta"))
    Bb = Bld(rm)
    Bb.mul([(fv_mulout(rm.X2, 4 * Q * Q), beta)], rm.X2)
    Eb = sched(Bb.seq)
    A(lbl("L_nobeta") + ":")
    # y -> K - y where the record says so
    Bn = Bld(rm)
    y_in = fv_mulout(rm.Y2, 4 * Q * Q)
    y_neg = Bn.neg(y_in, rm.T2)
    Bn.seq += [i_cnd(rm.Y2[j], rm.Y2[j], rm.T2[j], S_NEG) for j in range(NL)]
    sched(Bn.seq, pre={S_NEG: -1})
    y2 = FV(rm.Y2, max(y_in.lmax, y_neg.lmax), max(y_in.tmax, y_neg.tmax), max(y_in.vmax, y_neg.vmax))
    Ba = seq_madd(rm, y2=y2)
    Ea = sched(Ba.seq)
    # digit zero: keep the saved accumulator
    A("s_nop 1")
    for d, s_ in zip(rm.X1 + rm.Y1 + rm.Z1, rm.SX + rm.SY + rm.SZ):
        A("v_cndmask_b32_e64 %s, %s, %s, %s" % (d, s_, d, S_NZ))
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_lt_u32 %s, %d" % (S_STEP, N_STEPS))
    A("s_cbranch_scc1 " + lbl("L_step"))
    A("s_waitcnt vmcnt(0)")                                                   # the last prefetch is not used
    # epilogue: back to R = 2^256 (one multiplication by the plain integer 2^256 mod q), packed lazy words; flag = (Z == 0 mod q)
    c256 = const_to(A, R32 % Q)
    Bo = Bld(rm)
    Bo.mul([(fv_acc(rm.X1), c256)], rm.X1)
    Bo.mul([(fv_acc(rm.Y1), c256)], rm.Y1)
    Bo.mul([(fv_z(rm.Z1), c256)], rm.Z1)
    Bo.seq += seq_pack(rm.X1, rm.T0[:8]) + seq_pack(rm.Y1, rm.T1[:8]) + seq_pack(rm.Z1, rm.T2[:8])
    u = rm.X2
    Bo.seq += [i_or(u[0], rm.T2[0], rm.T2[1])] + [i_or(u[0], u[0], rm.T2[j]) for j in range(2, 8)] + [i_cmpz(S_M1, u[0])]
    Bo.seq += [i_xor(u[j], (Q >> (32 * j)) & M32, rm.T2[j]) for j in range(8)]
    Bo.seq += [i_or(u[0], u[0], u[j]) for j in range(1, 8)] + [i_cmpz(S_M2, u[0])]
    sched(Bo.seq)
    A("s_nop 1")
    A("s_or_b64 %s, %s, %s" % (S_EXC, S_M1, S_M2))
    st(rm.T0[:8], rm.tid96, "res", 0); st(rm.T1[:8], rm.tid96, "res", 32); st(rm.T2[:8], rm.tid96, "res", 64)
    A("v_cndmask_b32_e64 %s, 0, 1, %s" % (rm.flag, S_EXC))
    A("global_store_dword %s, %s, %%[exc]" % (rm.tid4, rm.flag))
    A("s_waitcnt vmcnt(0)")
    stats = dict(double=len(Ed.order), madd=len(Ea.order), beta=len(Eb.order), vgpr_end=rm.end,
                 mult_double=sum(1 for i in Bd.seq if i.op in ("mad", "mul_lo")), mult_madd=sum(1 for i in Ba.seq if i.op in ("mad", "mul_lo")),
                 mult_beta=sum(1 for i in Bb.seq if i.op in ("mad", "mul_lo")), mult_epilogue=sum(1 for i in Bo.seq if i.op in ("mad", "mul_lo")))
    return L, rm, stats


# ---- the table kernel -----------------------------------------------------------------------------------------------------------------
def emit_table():
    """Operands as g1_smul_table_asm: %[tid] %[poff] (VGPR), %[n] (SGPR), %[pts] %[jtab] %[tab] %[zc] (SGPR pairs).

    The 16 multiples T_e = (e + 1) P are built on the curve where P is affine (one doubling, 14 mixed additions), each with its own z_e; every
    step multiplies Z by a factor g_e (2Y for the doubling, 2H for an addition), so z_e = g_1 ... g_e and the LAST z is a common multiple of all:
    T_e rescaled to Z* = z_15 is (X_e c^2, Y_e c^3) with c = g_{e+1} ... g_15 -- ONE backward pass with a running product, no prefix
    products and no inversion.  Scratch (jtab) entry e = (X_e, Y_e, g_e), 27 words in eight 16-byte chunks, chunk-major: chunk k of entry e
    of lane t sits at ((8 e + k) n + t) 16, so every load / store instruction of a wave covers one contiguous kilobyte."""
    rm = RegMap(table_kernel=True)
    assert rm.end <= 168
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s_: "%s_%%=" % s_
    S_E, S_N16 = S_STEP, S_N4
    XY1, XY2 = rm.X1 + rm.Y1, rm.X2 + rm.Y2
    tid16 = rm.tid32                                                         # lane * 16; the zc offset (lane * 32) is rebuilt at the end

    def sched(seq):
        E = Emitter()
        E.schedule(seq)
        L.extend(E.lines)
        return E

    def chunks(regs):
        return [regs[i:i + 4] for i in range(0, len(regs), 4)]

    def jt_off(dst, sidx):                                                   # dst = sidx * (8 n 16) + lane * 16
        A("s_mul_i32 %s, %s, %s" % (S_TMP, sidx, S_NJT))
        A("v_add_u32_e32 %s, %s, %s" % (dst, S_TMP, tid16))

    def jt_io(kind, off, groups, first_chunk=0):
        """groups: register runs, each starting a new chunk; `off` is advanced chunk by chunk (and left past the last one)"""
        if first_chunk:
            A("s_mul_i32 %s, %s, %d" % (S_TMP, S_N16, first_chunk))
            A("v_add_u32_e32 %s, %s, %s" % (off, S_TMP, off))
        n = 0
        for g in groups:
            for c in chunks(g):
                if n:
                    A("v_add_u32_e32 %s, %s, %s" % (off, S_N16, off))
                sfx = {1: "dword", 2: "dwordx2", 3: "dwordx3", 4: "dwordx4"}[len(c)]
                if kind == "st":
                    A("global_store_%s %s, %s, %%[jtab]" % (sfx, off, vrange(c)))
                else:
                    A("global_load_%s %s, %s, %%[jtab]" % (sfx, vrange(c), off))
                n += 1
        if kind == "st":
            A("s_nop 0")

    def entry_off(dst, sidx, stride_s, tid_v):
        A("s_mul_i32 %s, %s, %s" % (S_TMP, sidx, stride_s))
        A("v_add_u32_e32 %s, %s, %s" % (dst, S_TMP, tid_v))

    def mov_const(regs, value):
        for j, l in enumerate(limbs29(value)):
            A("v_mov_b32_e32 %s, 0x%08x" % (regs[j], l))

    one = RP % Q
    prologue(A, rm)
    A("v_lshlrev_b32_e32 %s, 4, %%[tid]" % tid16)
    A("v_lshlrev_b32_e32 %s, 6, %%[tid]" % rm.tid64)
    A("s_lshl_b32 %s, %%[n], 6" % S_N64)
    A("s_lshl_b32 %s, %%[n], 4" % S_N16)
    A("s_lshl_b32 %s, %%[n], 7" % S_NJT)
    # ---- T_0 = P as the affine point (X, Y) of the curve scaled by its own Z, converted to 29-bit limbs / R' = 2^261 (x 2^266 / 2^261 = x 2^5)
    ld(rm.T1[:8], "%[poff]", "pts", 0); ld(rm.T2[:8], "%[poff]", "pts", 32)
    c266 = const_to(A, (1 << 266) % Q)
    mov_const(rm.Z1, one)
    A("s_waitcnt vmcnt(0)")
    Bc = Bld(rm)
    Bc.seq += seq_unpack(rm.T1[:8], rm.X1) + seq_unpack(rm.T2[:8], rm.Y1)
    Bc.mul([(FV(rm.X1, M29, (1 << 24) - 1, (1 << 256) - 1), c266)], rm.X1)     # any 256-bit input
    Bc.mul([(FV(rm.Y1, M29, (1 << 24) - 1, (1 << 256) - 1), c266)], rm.Y1)
    sched(Bc.seq)
    A("v_mov_b32_e32 %s, %s" % (rm.off, tid16))
    jt_io("st", rm.off, [XY1])
    for d, s_ in zip(rm.SY + rm.SZ, XY1):                                    # (x, y) of P stay in registers for the 14 additions
        A("v_mov_b32_e32 %s, %s" % (d, s_))
    Ed = sched(seq_double(rm).seq)                                           # T_1 = 2 P; g_1 = Z3 = 2 Y
    A("s_mov_b32 %s, 1" % S_E)
    jt_off(rm.off, S_E)
    jt_io("st", rm.off, [XY1, rm.Z1])
    # ---- T_e = T_{e-1} + P, e = 2 .. 15; g_e = 2 H.  No waits in this loop: the stores drain while the next addition runs.
    A("s_mov_b32 %s, 2" % S_E)
    A(lbl("T_build") + ":")
    for d, s_ in zip(XY2, rm.SY + rm.SZ):
        A("v_mov_b32_e32 %s, %s" % (d, s_))
    Ba = seq_madd(rm, h2_out=rm.SX)
    Ea = sched(Ba.seq)
    g_bounds = Ba.h2
    jt_off(rm.off, S_E)
    jt_io("st", rm.off, [XY1, rm.SX])
    A("s_add_u32 %s, %s, 1" % (S_E, S_E))
    A("s_cmp_lt_u32 %s, 16" % S_E)
    A("s_cbranch_scc1 " + lbl("T_build"))
    A("s_waitcnt vmcnt(0)")
    # ---- backward: c = g_{e+1} ... g_15 (c = 1 for e = 15); x' = X c^2, y' = Y c^3; packed stores; then c *= g_e.
    # Two entry buffers, the loop unrolled by two: entry e - 1 is requested before entry e is used, and the wait is COUNTED (vmcnt(8): the eight
    # loads just issued may stay in flight; everything older -- this entry's loads, the previous stores -- has returned: VMEM returns in order).
    bufs = [(rm.X2, rm.Y2, rm.SX), (rm.X1, rm.Y1, rm.SY)]
    m2 = lambda r: fv_mulout(r, 4 * Q * Q)
    gz = fv_z(rm.SX)

    def half(cur, nxt, may_be_last):
        Xc, Yc, Gc = cur
        if may_be_last:
            A("s_cmp_eq_u32 %s, 0" % S_E)
            A("s_cbranch_scc1 " + lbl("T_last"))
        A("s_sub_u32 %s, %s, 1" % (S_IDX, S_E))
        jt_off(rm.off, S_IDX)
        jt_io("ld", rm.off, [nxt[0] + nxt[1], nxt[2]])
        A("s_waitcnt vmcnt(8)")
        if may_be_last:
            A("s_branch " + lbl("T_go"))
            A(lbl("T_last") + ":")
            A("s_waitcnt vmcnt(0)")
            A(lbl("T_go") + ":")
        Bk = Bld(rm)
        c = m2(rm.T0)
        c2 = Bk.mul([(c, c)], rm.T1)                                           # c^2
        Bk.mul([(fv_acc(Xc), c2)], Xc)                                         # x'
        c3 = Bk.mul([(c2, c)], rm.T1)                                          # c^3
        Bk.mul([(fv_acc(Yc), c3)], Yc)                                         # y'
        g = FV(Gc, max(g_bounds.lmax, gz.lmax), max(g_bounds.tmax, gz.tmax), max(g_bounds.vmax, gz.vmax))     # 2 H of an addition, or Z3 of the doubling (entry 1)
        Bk.mul([(c, g)], rm.T0)                                                # c *= g_e  (entry 0 has no g: the product is not used)
        Bk.seq += seq_pack(Xc, rm.T1[:8]) + seq_pack(Yc, rm.T2[:8])
        sched(Bk.seq)
        entry_off(rm.off, S_E, S_N64, rm.tid64)
        st(rm.T1[:8], rm.off, "tab", 0); st(rm.T2[:8], rm.off, "tab", 32)

    mov_const(rm.T0, one)
    A("s_mov_b32 %s, 15" % S_E)
    jt_off(rm.off, S_E)
    jt_io("ld", rm.off, [bufs[0][0] + bufs[0][1], bufs[0][2]])
    A(lbl("T_back") + ":")
    half(bufs[0], bufs[1], False)                                              # e odd: e - 1 exists
    A("s_sub_u32 %s, %s, 1" % (S_E, S_E))
    half(bufs[1], bufs[0], True)                                               # e even: the last one is e = 0
    A("s_cmp_eq_u32 %s, 0" % S_E)
    A("s_cbranch_scc1 " + lbl("T_back_done"))
    A("s_sub_u32 %s, %s, 1" % (S_E, S_E))
    A("s_branch " + lbl("T_back"))
    A(lbl("T_back_done") + ":")
    # ---- total Z of the table on the ORIGINAL curve: Zt = Z* Z_P; blinding point and correction on the table's curve
    ld(rm.T2[:8], "%[poff]", "pts", 64)
    A("s_waitcnt vmcnt(0)")
    c266 = const_to(A, (1 << 266) % Q)
    Bz = Bld(rm)
    Bz.seq += seq_unpack(rm.T2[:8], rm.SX)
    zp = Bz.mul([(FV(rm.SX, M29, (1 << 24) - 1, (1 << 256) - 1), c266)], rm.SX)
    zt = Bz.mul([(fv_z(rm.Z1), zp)], rm.Z1)                                    # Zt
    zt2 = Bz.mul([(zt, zt)], rm.T0)                                            # Zt^2
    zt3 = Bz.mul([(zt2, zt)], rm.Y1)                                           # Zt^3
    sched(Bz.seq)
    t_ = int.from_bytes(hashlib.sha3_256(b"arkmpc g1 window-loop blinding point R0").digest(), "big") % RORD
    R0 = EC.g1_mul(EC.GEN, t_)
    C = EC.g1_mul(R0, (1 << (5 * (N_STEPS // 2 - 1))) % RORD)
    for idx, (cx, cy) in ((16, R0), (17, (C[0], Q - C[1]))):
        Bq = Bld(rm)
        kx = const_to(A, mont(cx))
        Bq.mul([(zt2, kx)], rm.X2)
        sched(Bq.seq)
        Bq = Bld(rm)
        ky = const_to(A, mont(cy))
        Bq.mul([(zt3, ky)], rm.Y2)
        Bq.seq += seq_pack(rm.X2, rm.T1[:8]) + seq_pack(rm.Y2, rm.T2[:8])
        sched(Bq.seq)
        A("s_mov_b32 %s, %d" % (S_IDX, idx))
        entry_off(rm.off, S_IDX, S_N64, rm.tid64)
        st(rm.T1[:8], rm.off, "tab", 0); st(rm.T2[:8], rm.off, "tab", 32)
    # Zt back to R = 2^256, packed (lazy range: the finish kernel multiplies it into the result's Z)
    c256 = const_to(A, R32 % Q)
    Bo = Bld(rm)
    Bo.mul([(zt, c256)], rm.Z1)
    Bo.seq += seq_pack(rm.Z1, rm.X2[:8])
    sched(Bo.seq)
    A("v_lshlrev_b32_e32 %s, 1, %s" % (rm.off, tid16))
    st(rm.X2[:8], rm.off, "zc", 0)
    A("s_waitcnt vmcnt(0)")
    return L, rm, dict(double=len(Ed.order), madd=len(Ea.order), vgpr_end=rm.end)


def emit_msm_acc():
    """Bucket accumulation of the variable-base MSM and the fixed-base chain (as g1_msm_acc_asm of tools/gen_ec_asm.py) on the 29-bit mixed
    addition.  Members are 64-byte affine records whose coordinates are ALREADY in this representation's Montgomery form (x 2^261 mod q as a
    256-bit integer: the producers multiply by 2^5 once); the sum leaves in the engine's format (R = 2^256, lazy words).  The next member's
    coordinates are requested BEFORE the addition of the current one (they land in registers the addition does not touch).  A chain that
    meets H = 0 ends with Z = 0 (mod q): one test in the epilogue raises the lane's flag.
    Operands: %[lo4] %[len] (VGPR), %[maxlen] (SGPR), %[vals] %[aff] %[exc] (SGPR pairs), %[dst] (VGPR pair), %[t4] (VGPR)."""
    rm = RegMap(msm=True)
    L = []
    A = L.append
    ld, st = mem_ops(A)
    lbl = lambda s: "%s_%%=" % s

    def sched(seq, pre=None):
        E = Emitter()
        if pre: E.lastw.update(pre)
        E.schedule(seq)
        L.extend(E.lines)
        return E

    def load_xy():
        A("v_lshlrev_b32_e32 %s, 6, %s" % (rm.off, rm.rec))                 # 64 bytes per member; the shift drops the sign bit (index < 2^25)
        ld(rm.LD1, rm.off, "aff", 0); ld(rm.LD2, rm.off, "aff", 32)

    prologue(A, rm)
    A("v_add_u32_e32 %s, -1, %%[len]" % rm.last)
    A("global_load_dword %s, %%[lo4], %%[vals]" % rm.rec)
    A("v_min_u32_e32 %s, 1, %s" % (rm.tmp, rm.last))
    A("v_lshl_add_u32 %s, %s, 2, %%[lo4]" % (rm.tmp, rm.tmp))
    A("global_load_dword %s, %s, %%[vals]" % (rm.rec2, rm.tmp))
    A("s_waitcnt vmcnt(1)")
    load_xy()
    A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
    for j, l in enumerate(limbs29(RP % Q)):
        A("v_mov_b32_e32 %s, 0x%08x" % (rm.Z1[j], l))
    A("s_waitcnt vmcnt(0)")
    # accumulator = +- first member: unpack, negate y where the sign bit says so, one carry pass (the accumulator invariant wants normalised limbs)
    B0 = Bld(rm)
    B0.seq += seq_unpack(rm.LD1, rm.X1) + seq_unpack(rm.LD2, rm.Y1)
    y_in = fv_mulout(rm.Y1, 4 * Q * Q)
    y_neg = B0.neg(y_in, rm.T2)
    B0.seq += [i_cnd(rm.Y1[j], rm.Y1[j], rm.T2[j], S_NEG) for j in range(NL)]
    y0 = B0.norm(FV(rm.Y1, max(y_in.lmax, y_neg.lmax), max(y_in.tmax, y_neg.tmax), max(y_in.vmax, y_neg.vmax)))
    check_acc(y0, "first member")
    sched(B0.seq, pre={S_NEG: -1})
    A("v_mov_b32_e32 %s, %s" % (rm.rec, rm.rec2))
    load_xy()
    A("s_mov_b32 %s, 1" % S_STEP)
    A("s_cmp_ge_u32 %s, %%[maxlen]" % S_STEP)
    A("s_cbranch_scc1 " + lbl("L_done"))
    EC.align_head(A)
    A(lbl("L_step") + ":")
    A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
    A("v_min_u32_e32 %s, %s, %s" % (rm.tmp, S_TMP, rm.last))
    A("v_lshl_add_u32 %s, %s, 2, %%[lo4]" % (rm.tmp, rm.tmp))
    A("global_load_dword %s, %s, %%[vals]" % (rm.rec2, rm.tmp))
    A("v_cmp_gt_u32_e64 %s, %%[len], %s" % (S_NZ, S_STEP))                  # this lane still has a member at this step
    A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
    for d, s_ in zip(rm.SX + rm.SY + rm.SZ, rm.X1 + rm.Y1 + rm.Z1):
        A("v_mov_b32_e32 %s, %s" % (d, s_))
    A("s_waitcnt vmcnt(1)")
    Bn = Bld(rm)
    Bn.seq += seq_unpack(rm.LD1, rm.X2) + seq_unpack(rm.LD2, rm.Y2)
    y_in = fv_mulout(rm.Y2, 4 * Q * Q)
    y_neg = Bn.neg(y_in, rm.T2)
    Bn.seq += [i_cnd(rm.Y2[j], rm.Y2[j], rm.T2[j], S_NEG) for j in range(NL)]
    sched(Bn.seq, pre={S_NEG: -1})
    y2 = FV(rm.Y2, max(y_in.lmax, y_neg.lmax), max(y_in.tmax, y_neg.tmax), max(y_in.vmax, y_neg.vmax))
    # the member after this one: its coordinates fly during the addition
    A("s_waitcnt vmcnt(0)")
    A("v_mov_b32_e32 %s, %s" % (rm.rec, rm.rec2))
    load_xy()
    Ba = seq_madd(rm, y2=y2)
    Ea = sched(Ba.seq)
    A("s_nop 1")
    for d, s_ in zip(rm.X1 + rm.Y1 + rm.Z1, rm.SX + rm.SY + rm.SZ):
        A("v_cndmask_b32_e64 %s, %s, %s, %s" % (d, s_, d, S_NZ))
    A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
    A("s_cmp_lt_u32 %s, %%[maxlen]" % S_STEP)
    A("s_cbranch_scc1 " + lbl("L_step"))
    A(lbl("L_done") + ":")
    A("s_waitcnt vmcnt(0)")                                                   # the last prefetch (clamped to the last member) is not used
    c256 = const_to(A, R32 % Q)
    Bo = Bld(rm)
    Bo.mul([(fv_acc(rm.X1), c256)], rm.X1)
    Bo.mul([(fv_acc(rm.Y1), c256)], rm.Y1)
    Bo.mul([(fv_z(rm.Z1), c256)], rm.Z1)
    Bo.seq += seq_pack(rm.X1, rm.T0[:8]) + seq_pack(rm.Y1, rm.T1[:8]) + seq_pack(rm.Z1, rm.T2[:8])
    u = rm.X2
    Bo.seq += [i_or(u[0], rm.T2[0], rm.T2[1])] + [i_or(u[0], u[0], rm.T2[j]) for j in range(2, 8)] + [i_cmpz(S_M1, u[0])]
    Bo.seq += [i_xor(u[j], (Q >> (32 * j)) & M32, rm.T2[j]) for j in range(8)]
    Bo.seq += [i_or(u[0], u[0], u[j]) for j in range(1, 8)] + [i_cmpz(S_M2, u[0])]
    sched(Bo.seq)
    A("s_nop 1")
    A("s_or_b64 %s, %s, %s" % (S_EXC, S_M1, S_M2))
    for k, regs in enumerate((rm.T0[:4], rm.T0[4:8], rm.T1[:4], rm.T1[4:8], rm.T2[:4], rm.T2[4:8])):
        A("global_store_dwordx4 %%[dst], %s, off offset:%d" % (vrange(regs), 16 * k))
    A("s_nop 0")
    A("v_cndmask_b32_e64 %s, 0, 1, %s" % (rm.flag, S_EXC))
    A("global_store_dword %%[t4], %s, %%[exc]" % rm.flag)
    A("s_waitcnt vmcnt(0)")
    return L, rm, dict(madd=len(Ea.order), vgpr_end=rm.end)


def count_mults():
    rm = RegMap()
    cnt = lambda B: sum(1 for i in B.seq if i.op in ("mad", "mul_lo"))
    Bm = Bld(rm); Bm.mul([(fv_z(rm.X1), fv_z(rm.Y1))], rm.Z1)
    a = fv_z(rm.X1)
    Bs = Bld(rm); Bs.mul([(a, a)], rm.Z1)
    return dict(mul=cnt(Bm), sqr=cnt(Bs), double=cnt(seq_double(rm)), madd=cnt(seq_madd(rm)))


def emit_header(path):
    selftest(trials=24)
    lines, rm, st = emit_loop()
    cm = count_mults()
    n_dbl = 5 * (N_STEPS // 2 - 1)
    loop_m = n_dbl * cm["double"] + N_STEPS * cm["madd"] + (N_STEPS // 2) * cm["mul"] + 3 * cm["mul"]
    table_m = 2 * cm["mul"] + cm["double"] + 14 * cm["madd"] + 16 * (4 * cm["mul"] + cm["sqr"]) + (3 * cm["mul"] + cm["sqr"]) + 4 * cm["mul"] + cm["mul"]
    out = []
    out.append("// GENERATED by tools/gen_ec29_asm.py -- do not edit.  The BN254 G1 window loop and its table kernel on NINE 29-bit limbs (Montgomery radix 2^261):")
    out.append("// product-scanning multiplier with one 64-bit column accumulator, carry-free limb additions, bounds carried by the generator.")
    out.append("// double: %d instructions, mixed add: %d, VGPRs v%d..v%d." % (st["double"], st["madd"], rm.first, rm.end - 1))
    out.append("// multiplier instructions per scalar-mul: window loop %d (%d doublings x %d, %d mixed additions x %d, %d beta + 3 conversion products x %d), table %d"
               % (loop_m, n_dbl, cm["double"], N_STEPS, cm["madd"], N_STEPS // 2, cm["mul"], table_m))
    out.append("#pragma once")
    out.append("#define G1_ASM29_MULT_INSTRS_LOOP %d\n#define G1_ASM29_MULT_INSTRS_TABLE %d" % (loop_m, table_m))
    out.append("#define G1_ASM29_JT_STRIDE %d" % JT_STRIDE)
    w32 = lambda v: ", ".join("0x%08xu" % ((v >> (32 * i)) & M32) for i in range(8))
    out.append("// member records of g1_msm_acc29_asm hold x 2^261 (this representation's Montgomery form) where the engine's records hold x 2^256:")
    out.append("// FQ_MUL(record, TO29) converts there (x 2^256 * 2^261 / 2^256), FQ_MUL(record29, FROM29) back (x 2^261 * 2^251 / 2^256)")
    out.append("__device__ constexpr u32 G1_ASM29_TO29[8] = {%s};" % w32((1 << 261) % Q))
    out.append("__device__ constexpr u32 G1_ASM29_FROM29[8] = {%s};" % w32((1 << 251) % Q))
    out.append("__device__ __forceinline__ void g1_smul_loop29_asm(u32 tid, u32 n, u32 ptid, u32 np, const u64* tab, const u32* dig, u64* res, u32* exc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(lines))
    out.append("        :")
    out.append('        : [tid] "v"(tid), [n] "s"(n), [ptid] "v"(ptid), [np] "s"(np), [tab] "s"(tab), [dig] "s"(dig), [res] "s"(res), [exc] "s"(exc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(rm.first, rm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    tlines, trm, tst = emit_table()
    out.append("// the window table (16 multiples on the curve where P is affine, rescaled to a common Z; blinding point and correction): %d asm lines, VGPRs v%d..v%d"
               % (len(tlines), trm.first, trm.end - 1))
    out.append("__device__ __forceinline__ void g1_smul_table29_asm(u32 tid, u32 poff, u32 n, const u64* pts, u64* jtab, u64* tab, u64* zc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(tlines))
    out.append("        :")
    out.append('        : [tid] "v"(tid), [poff] "v"(poff), [n] "s"(n), [pts] "s"(pts), [jtab] "s"(jtab), [tab] "s"(tab), [zc] "s"(zc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(trm.first, trm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    mlines, mrm, mst = emit_msm_acc()
    out.append("// bucket accumulation of the variable-base MSM / the fixed-base chain on the same mixed addition: %d asm lines, VGPRs v%d..v%d" % (len(mlines), mrm.first, mrm.end - 1))
    out.append("__device__ __forceinline__ void g1_msm_acc29_asm(u32 lo4, u32 len, u32 maxlen, const u32* vals, const u64* aff, u64* dst, u32 t4, u32* exc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(mlines))
    out.append("        :")
    out.append('        : [lo4] "v"(lo4), [len] "v"(len), [maxlen] "s"(maxlen), [vals] "s"(vals), [aff] "s"(aff), [dst] "v"(dst), [t4] "v"(t4), [exc] "s"(exc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mrm.first, mrm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    import json
    with open(os.path.join(os.path.dirname(path), "ec29_asm_stats.json"), "w") as f:
        json.dump({"mult_instrs_loop": loop_m, "mult_instrs_table": table_m, "mult_instrs_per_mul": cm["mul"], "mult_instrs_per_sqr": cm["sqr"],
                   "mult_instrs_double": cm["double"], "mult_instrs_madd": cm["madd"], "doublings": n_dbl, "mixed_additions": N_STEPS,
                   "double_body_instrs": st["double"], "madd_body_instrs": st["madd"]}, f, indent=1)
    return st, len(lines) + len(tlines) + len(mlines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "ec29_asm_kernels.inc"))
    a = ap.parse_args()
    if a.selftest:
        Ed, Ea = selftest(trials=200)
        print("multiplications at their operand bounds: %d distinct, ok" % selftest_extremes())
        print("double ok: %d instructions; madd ok: %d instructions" % (len(Ed.order), len(Ea.order)))
        sys.exit(0)
    st, nlines = emit_header(a.o)
    print("ec29: %d asm lines; %s" % (nlines, st))
