#!/usr/bin/env python3
"""Generator for the hand-scheduled Curve25519 (twisted Edwards, a = -1) window loop (ark-mpc_amd/csrc/ed_asm_kernels.inc) -- the
Edwards counterpart of tools/gen_ec_asm.py, built from the same multiplier rows, scheduler and emulator.

What differs from the BN254 loop:
  * the field is 2^255 - 19, and the loop does NOT use Montgomery multiplication: a product is the plain 512-bit product (8 rows of
    v_mad_u64_u32 into 16 fixed column registers), folded with 2^256 = 38 (8 more multiplier instructions) and 2^255 = 19 -- 72 multiplier
    instructions instead of the 136 of a Montgomery block, 44 for a squaring (doubled-operand rows).  The arkworks coordinates the
    boundary hands over are in Montgomery form, x R; extended twisted-Edwards coordinates are projective, so (X R : Y R : Z R : T R)
    read as plain field elements is the SAME point (and so is every cached table entry, which is linear in them but for the
    constant 2d, already multiplied in by the prep kernel), and the plain result (X : Y : Z : T) read back as Montgomery-form limbs is
    again the same point: no conversion on either side.  Results are compared on affine coordinates, like every point result;
  * values live in [0, 2^255 + 1463): a product is < 2^255 + 19 * 77, an addition is a 257-bit sum folded with 19 * (bits 255..256), a
    subtraction is a + (2q - b) folded the same way -- both stay below 2^255 + 57;
  * the group law is COMPLETE (add-2008-hwcd-3 with a = -1 on a curve where -1 is a square and d is not): no exceptional lanes, no
    blinding point, no flags -- the accumulator starts at the identity and a zero digit adds the identity's table entry;
  * table entries are cached extended points (Y+X, Y-X, 2dT, 2Z): 7 multiplications per addition plus one for T when the next
    operation needs it; doubling is dbl-2008-hwcd rearranged to avoid negations (3S + 4M, + 1 for T).

One step = 5 doublings (the last one with T) + one addition of +-T[|d|]; 51 signed 5-bit windows cover the 253-bit scalars.
The two bodies are executed by the single-lane emulator against the affine Edwards law in Python integers (--selftest).

Reference semantics: CurvePoint * Scalar on ark_curve25519::EdwardsProjective (online-phase/src/algebra/curve/curve.rs:403-409,
README.md:24), PointShare * Scalar (curve/share.rs:108-114).
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_asm_kernels as G
import gen_ec_asm as EC
from gen_asm_kernels import Ins, Emitter, M32, R, i_mov, i_mad, i_addco, i_addc, i_subco, i_subb, i_cnd, regs_of

Q = dict(G.FIELDS)["CURVE25519_FQ"]
L_ORD = dict(G.FIELDS)["CURVE25519_FR"]
B255 = 1 << 255
LIM = B255 + 19 * 77                 # every value the loop holds is below this (a folded product: top <= 2 * 38 + 1)
D_ED = (-121665 * pow(121666, -1, Q)) % Q

S_JUNK, S_CY2, S_INV = EC.S_JUNK, EC.S_CY2, EC.S_INV
S_P = EC.S_P
S_STEP, S_N4, S_N128, S_TMP, S_DBL = "s37", "s38", "s39", "s40", "s41"
S_NEG, S_CY3, S_CY4 = "s[44:45]", EC.S_CY3, EC.S_CY4
CLOBBER_SGPRS = ["s%d" % i for i in range(16, 56)]
N_WINDOWS = 51
N_TABLE = 17                         # identity + 16 multiples

_POOL = [("vcc", S_CY2), (S_CY3, S_CY4)]
_turn = [0]


def _carries():
    _turn[0] ^= 1
    return _POOL[_turn[0]]


class RegMap:
    def __init__(self, first=8):
        rg = G.Regs(first)
        self.TWOQ = rg.vec(8)
        self.X1, self.Y1, self.Z1, self.T1 = (rg.vec(8, 4) for _ in range(4))      # accumulator, extended coordinates
        self.QP, self.QM, self.QT, self.QZ = (rg.vec(8, 4) for _ in range(4))      # table entry: Y+X, Y-X, 2dT, 2Z
        self.A, self.Bv, self.Cv = (rg.vec(8, 4) for _ in range(3))                # temporaries
        self.D2 = rg.vec(8, 4)                                                      # 2a for the squaring rows
        self.C = [rg.pair() for _ in range(16)]                                     # product columns: (limb, 0) pairs, high halves stay 0
        self.q = [rg.pair() for _ in range(8)]
        self.m = rg.one()
        self.c9, self.top = rg.one(), rg.one()
        self.rec, self.off, self.tid4, self.tid128, self.tmp = (rg.one() for _ in range(5))
        self.first, self.end = first, rg.next


def i_and(d, a, b): return G.i_and(d, a, b)
def i_lshr(d, a, sh): return EC.i_lshr(d, a, sh)
def i_shl(d, a, sh): return EC.i_shl(d, a, sh)
def i_or(d, a, b): return EC.i_or(d, a, b)
def i_mul24(d, k, a): return Ins("v_mul_u32_u24_e32 %s, %d, %s" % (d, k, a), "mul24", (d, k, a), rd=regs_of(a), wr=[d])


def fold(rm, s, c9, out, cy):
    """out = (s mod 2^255) + 19 * (bits 255.. of the 257-bit value c9 * 2^256 + s).  For a value < 4 * 2^255 the result is < 2^255 + 57;
    for the sums this generator forms it is < 2^255 + 19 (see the module docstring).  c9 = None: the value has no 257th bit."""
    seq = [i_lshr(rm.top, s[7], 31)]
    if c9 is not None:
        seq += [i_shl(rm.c9, c9, 1), i_or(rm.top, rm.top, rm.c9)]
    seq += [i_mul24(rm.top, 19, rm.top), i_and(rm.c9, 0x7fffffff, s[7])]
    seq += [i_addco(out[0], s[0], rm.top, cy)] + [i_addc(out[j], 0, s[j], cy) for j in range(1, 7)] + [i_addc(out[7], 0, rm.c9, cy)]
    return seq


def _reduce(rm, out, cy):
    """out = (C[0..7] + 38 C[8..15]) folded below 2^255 + 19 * 77: 2^256 = 38 and 2^255 = 19 (mod q)"""
    C, q = rm.C, rm.q
    seq = [i_mad(q[j], C[8 + j][0], 38, C[j]) for j in range(8)]
    seq += [i_mov(out[0], q[0][0]), i_addco(out[1], q[1][0], q[0][1], cy)]
    seq += [i_addc(out[j], q[j][0], q[j - 1][1], cy) for j in range(2, 8)]
    seq += [i_addc(rm.m, 0, q[7][1], cy)]                                         # ninth limb <= 38
    return seq + fold(rm, out, rm.m, out, cy)


_row = [0]
def _rowcy():
    _row[0] ^= 1
    return "vcc" if _row[0] else S_CY2


def pmul(rm, a, b, out):
    """out = a b mod q (plain product, no Montgomery factor), out < 2^255 + 19 * 77, for a, b < 2^256; out may alias an operand.
    Row r adds a * b_r into the fixed columns C[r .. r + 8]."""
    C, q = rm.C, rm.q
    seq = []
    for r in range(8):
        cy = _rowcy()
        seq += [i_mad(q[j], a[j], b[r], 0 if r == 0 else C[r + j]) for j in range(8)]
        seq += [i_mov(C[r][0], q[0][0]), i_addco(C[r + 1][0], q[1][0], q[0][1], cy)]
        seq += [i_addc(C[r + j][0], q[j][0], q[j - 1][1], cy) for j in range(2, 8)]
        seq += [i_addc(C[r + 8][0], 0, q[7][1], cy)]
    return seq + _reduce(rm, out, _rowcy())


def psqr(rm, a, out):
    """out = a^2 mod q, out < 2^255 + 19 * 77.  a is first folded below 2^255 IN PLACE (the same residue; callers keep using it), so
    that d = 2a fits eight limbs; row r then multiplies a_r only into the columns from 2r up -- a_r a_r at column 2r, d_j a_r at
    column r + j for j > r (the low bit of d_{r+1} is the top bit of a_r, which belongs to 2 a_r: masked off) -- 36 products."""
    C, q, d, m = rm.C, rm.q, rm.D2, rm.m
    seq = fold(rm, a, None, a, _rowcy())
    seq += [i_shl(d[0], a[0], 1)] + [EC.i_alignbit(d[j], a[j], a[j - 1], 31) for j in range(1, 8)]
    for r in range(8):
        cy = _rowcy()
        if r < 7:
            seq += [i_and(m, -2 & M32, d[r + 1])]
        seq += [i_mad(q[j], a[r] if j == r else (m if j == r + 1 else d[j]), a[r], 0 if r == 0 else C[r + j]) for j in range(r, 8)]
        seq += [i_mov(C[2 * r][0], q[r][0])]
        if r < 7:
            seq += [i_addco(C[2 * r + 1][0], q[r + 1][0], q[r][1], cy)]
            seq += [i_addc(C[r + j][0], q[j][0], q[j - 1][1], cy) for j in range(r + 2, 8)]
            seq += [i_addc(C[r + 8][0], 0, q[7][1], cy)]
        else:
            seq += [i_mov(C[15][0], q[7][1])]
    return seq + _reduce(rm, out, _rowcy())


def add_lz(rm, a, b, out, tmp):
    c1, c2 = _carries()
    seq = [i_addco(tmp[0], a[0], b[0], c1)] + [i_addc(tmp[j], a[j], b[j], c1) for j in range(1, 8)]
    seq += [i_addc(rm.c9, 0, rm.C[0][1], c1)]            # the 257th bit (Tz[8][1] holds 0; src1 must be a VGPR)
    return seq + fold(rm, tmp, rm.c9, out, c2)


def sub_lz(rm, a, b, out, tmp):
    """a - b = a + (2q - b) folded; b < 2^255 + 19 < 2q so the inner difference never borrows"""
    c1, c2 = _carries()
    seq = [i_subco(tmp[0], rm.TWOQ[0], b[0], c1)] + [i_subb(tmp[j], rm.TWOQ[j], b[j], c1) for j in range(1, 8)]
    seq += [i_addco(tmp[0], a[0], tmp[0], c2)] + [i_addc(tmp[j], a[j], tmp[j], c2) for j in range(1, 8)]
    seq += [i_addc(rm.c9, 0, rm.C[0][1], c2)]
    return seq + fold(rm, tmp, rm.c9, out, c1)


def seq_double(rm, with_t):
    """dbl-2008-hwcd with a = -1, arranged without negations: A = X^2, B = Y^2, Cc = 2 Z^2, E = 2 X Y, G = B - A, F' = Cc - G (= -F),
    Hn = A + B (= -H); (X3, Y3, Z3, T3) = (E F', G Hn, F' G, E Hn) is the standard result scaled by -1 throughout, i.e. the same point.
    In place on the accumulator; scratch: the table-entry registers and A / Bv / Cv."""
    X, Y, Z, T = rm.X1, rm.Y1, rm.Z1, rm.T1
    A, B, Cc, E, Gv, Fp, Hn, t = rm.A, rm.Bv, rm.QP, rm.QM, rm.QT, rm.QZ, rm.Cv, rm.T1
    s = []
    s += psqr(rm, X, A)
    s += psqr(rm, Y, B)
    s += pmul(rm, X, Y, E)
    s += psqr(rm, Z, Cc)
    s += add_lz(rm, E, E, E, t)                  # E = 2 X Y           (T1 is dead inside a doubling: scratch)
    s += add_lz(rm, Cc, Cc, Cc, t)               # Cc = 2 Z^2
    s += sub_lz(rm, B, A, Gv, t)                 # G = B - A
    s += add_lz(rm, A, B, Hn, t)                 # Hn = A + B
    s += sub_lz(rm, Cc, Gv, Fp, t)               # F' = Cc - G
    s += pmul(rm, E, Fp, X)
    s += pmul(rm, Gv, Hn, Y)
    s += pmul(rm, Fp, Gv, Z)
    if with_t:
        s += pmul(rm, E, Hn, T)
    return s


def seq_add(rm, with_t):
    """add-2008-hwcd-3 (a = -1) with a cached second operand (Y2+X2, Y2-X2, 2d T2, 2 Z2): complete on this curve.  In place."""
    X, Y, Z, T = rm.X1, rm.Y1, rm.Z1, rm.T1
    A, B, t = rm.A, rm.Bv, rm.Cv
    s = []
    s += sub_lz(rm, Y, X, A, t)                  # Y1 - X1
    s += add_lz(rm, Y, X, B, t)                  # Y1 + X1
    s += pmul(rm, A, rm.QM, A)                # A
    s += pmul(rm, B, rm.QP, B)                # B
    s += pmul(rm, T, rm.QT, rm.QT)            # C
    s += pmul(rm, Z, rm.QZ, rm.QZ)            # D
    E, H, F, Gv = rm.QP, rm.QM, X, Y             # X1, Y1 are dead now
    s += sub_lz(rm, B, A, E, t)                  # E = B - A
    s += add_lz(rm, B, A, H, t)                  # H = B + A
    s += sub_lz(rm, rm.QZ, rm.QT, F, t)          # F = D - C
    s += add_lz(rm, rm.QZ, rm.QT, Gv, t)         # G = D + C
    s += pmul(rm, F, Gv, Z)                   # Z3 = F G
    if with_t:
        s += pmul(rm, E, H, T)                # T3 = E H
    s += pmul(rm, E, F, X)                    # X3 = E F     (F lives in X1: read before the write by the row structure)
    s += pmul(rm, Gv, H, Y)                   # Y3 = G H
    return s


# ---- emulator -------------------------------------------------------------------------------------------------------------
class EdEmu(EC.EcEmu):
    def run(self, order):
        for ins in order:
            if ins.op == "mul24":
                a = ins.args
                self.v[a[0]] = (a[1] * (self.rd(a[2]) & 0xffffff)) & M32
            else:
                EC.EcEmu.run(self, [ins])


def _with_globals(fn):
    saved = (G.JUNK, G.S_INV, G.CY2)
    G.JUNK, G.S_INV, G.CY2 = S_JUNK, S_INV, S_CY2
    _row[0] = _turn[0] = 0                    # the carry-register rotation starts from the same state for every build: reproducible output
    try:
        return fn()
    finally:
        G.JUNK, G.S_INV, G.CY2 = saved


def ed_add_aff(p, q_):
    (x1, y1), (x2, y2) = p, q_
    k = D_ED * x1 * x2 * y1 * y2 % Q
    return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, Q) % Q, (y1 * y2 + x1 * x2) * pow(1 - k, -1, Q) % Q)


def ed_mul_aff(p, k):
    r = (0, 1)
    while k:
        if k & 1: r = ed_add_aff(r, p)
        p = ed_add_aff(p, p); k >>= 1
    return r


ED_BY = 4 * pow(5, -1, Q) % Q
def _bx():
    u = (ED_BY * ED_BY - 1) * pow(D_ED * ED_BY * ED_BY + 1, -1, Q) % Q
    x = pow(u, (Q + 3) // 8, Q)
    if (x * x - u) % Q: x = x * pow(2, (Q - 1) // 4, Q) % Q
    assert (x * x - u) % Q == 0
    return x if x % 2 == 0 else Q - x
ED_B = (_bx(), ED_BY)
mont = lambda v: v * R % Q
unmont = lambda v: v * pow(R, -1, Q) % Q


def _emu(rm):
    em = EdEmu()
    em.setv(rm.TWOQ, 2 * Q)
    for t in rm.C:
        em.v[t[1]] = 0
    for j in range(8):
        em.s[S_P[j]] = (Q >> (32 * j)) & M32
    em.s[S_INV] = (-pow(Q, -1, 1 << 32)) & M32
    return em


def _lazy(rng, v):
    """a representative of v mod q below the loop's bound"""
    return v + Q if (rng.random() < 0.5 and v + Q < LIM) else v


def selftest_field(trials=60, seed=5):
    """pmul / psqr / add_lz / sub_lz on the whole value range [0, 2^255 + 19 * 77), extremes included"""
    rng = random.Random(seed)
    edge = [0, 1, 19, Q - 1, Q, Q + 1, B255 - 1, B255, B255 + 18, B255 + 19, LIM - 1, M32, (1 << 224) - 1, B255 + 1462]
    def build(fn):
        def go():
            rm = RegMap()
            E = Emitter(); E.schedule(fn(rm))
            return E, rm
        return _with_globals(go)
    cases = {"mul": (lambda rm: pmul(rm, rm.X1, rm.Y1, rm.Z1), lambda a, b: a * b),
             "mul_alias": (lambda rm: pmul(rm, rm.X1, rm.Y1, rm.X1), lambda a, b: a * b),
             "sqr": (lambda rm: psqr(rm, rm.X1, rm.Z1), lambda a, b: a * a),
             "sqr_alias": (lambda rm: psqr(rm, rm.X1, rm.X1), lambda a, b: a * a),
             "add": (lambda rm: add_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.A), lambda a, b: a + b),
             "sub": (lambda rm: sub_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.A), lambda a, b: a - b)}
    for name, (fn, ref) in cases.items():
        E, rm = build(fn)
        outreg = rm.X1 if name.endswith("alias") else rm.Z1
        vals = [(a, b) for a in edge for b in edge] + [(rng.randrange(LIM), rng.randrange(LIM)) for _ in range(trials)]
        for a, b in vals:
            em = _emu(rm)
            em.setv(rm.X1, a); em.setv(rm.Y1, b)
            em.run(E.order)
            got = em.getv(outreg)
            assert got < LIM and got % Q == ref(a, b) % Q, (name, hex(a), hex(b), hex(got))
            if name == "sqr":                       # the operand was folded in place: same residue, still in range
                assert em.getv(rm.X1) % Q == a % Q and em.getv(rm.X1) < B255
    # the cached form of an accumulator (table kernel): 2d comes from SGPRs
    def gc():
        rm = RegMap()
        E = Emitter(); E.schedule(cached_seq(rm, S_P))
        return E, rm
    E, rm = _with_globals(gc)
    d2 = 2 * D_ED % Q
    for t in range(12):
        vals = [rng.choice(edge) if t < 4 else rng.randrange(LIM) for _ in range(4)]
        em = _emu(rm)
        for j in range(8):
            em.s[S_P[j]] = (d2 >> (32 * j)) & M32
        for regs, v in zip((rm.X1, rm.Y1, rm.Z1, rm.T1), vals):
            em.setv(regs, v)
        em.run(E.order)
        x, y, z, t_ = vals
        for regs, want in ((rm.A, y + x), (rm.Bv, y - x), (rm.Cv, d2 * t_), (rm.QZ, 2 * z)):
            assert em.getv(regs) < LIM and em.getv(regs) % Q == want % Q
    return True


def selftest(trials=30, seed=11):
    selftest_field()
    rng = random.Random(seed)
    def body(fn, with_t):
        def go():
            rm = RegMap()
            E = Emitter(); E.schedule(fn(rm, with_t))
            return E, rm
        return _with_globals(go)
    out = {}
    for with_t in (False, True):
        Ed, rm = body(seq_double, with_t)
        Ea, rm2 = body(seq_add, with_t)
        out[with_t] = (Ed, Ea)
        for t in range(trials):
            P = ed_mul_aff(ED_B, rng.randrange(1, L_ORD)) if t % 7 else (0, 1)
            z = rng.randrange(1, Q)
            ext = lambda p_, z_: (p_[0] * z_ % Q, p_[1] * z_ % Q, z_, p_[0] * p_[1] * z_ % Q)
            X, Y, Z, T = ext(P, z)
            em = _emu(rm)
            for regs, v in ((rm.X1, X), (rm.Y1, Y), (rm.Z1, Z), (rm.T1, T)):
                em.setv(regs, _lazy(rng, v))              # plain field elements, any representative below the loop's bound
            em.run(Ed.order)
            gx, gy, gz, gt = (em.getv(r_) % Q for r_ in (rm.X1, rm.Y1, rm.Z1, rm.T1))
            assert max(em.getv(r_) for r_ in (rm.X1, rm.Y1, rm.Z1)) < LIM
            zi = pow(gz, -1, Q)
            want = ed_add_aff(P, P)
            assert (gx * zi % Q, gy * zi % Q) == want, ("double", with_t, t)
            if with_t:
                assert gt * zi % Q == want[0] * want[1] % Q
            # addition with a cached operand, incl. the identity entry, the same point and the opposite point (complete law)
            kind = t % 6
            Qp = (0, 1) if kind == 0 else (P if kind == 1 else ((Q - P[0]) % Q, P[1]) if kind == 2 else ed_mul_aff(ED_B, rng.randrange(1, L_ORD)))
            z2 = rng.randrange(1, Q)
            X2, Y2, Z2, T2 = ext(Qp, z2)
            em = _emu(rm2)
            for regs, v in ((rm2.X1, X), (rm2.Y1, Y), (rm2.Z1, Z), (rm2.T1, T)):
                em.setv(regs, _lazy(rng, v))
            for regs, v in ((rm2.QP, (Y2 + X2) % Q), (rm2.QM, (Y2 - X2) % Q), (rm2.QT, 2 * D_ED * T2 % Q), (rm2.QZ, 2 * Z2 % Q)):
                em.setv(regs, _lazy(rng, v))
            em.run(Ea.order)
            gx, gy, gz, gt = (em.getv(r_) % Q for r_ in (rm2.X1, rm2.Y1, rm2.Z1, rm2.T1))
            assert max(em.getv(r_) for r_ in (rm2.X1, rm2.Y1, rm2.Z1)) < LIM
            zi = pow(gz, -1, Q)
            want = ed_add_aff(P, Qp)
            assert (gx * zi % Q, gy * zi % Q) == want, ("add", with_t, t, kind)
            if with_t:
                assert gt * zi % Q == want[0] * want[1] % Q
    # Niels addition (fixed-base chain): affine second operand, incl. the identity (1, 1, 0), the same point and the opposite point
    def gn():
        rm = RegMap()
        E = Emitter(); E.schedule(seq_add_niels(rm))
        return E, rm
    En_, rmn = _with_globals(gn)
    for t in range(trials):
        P = ed_mul_aff(ED_B, rng.randrange(1, L_ORD)) if t % 7 else (0, 1)
        z = rng.randrange(1, Q)
        X, Y, Z, T = (P[0] * z % Q, P[1] * z % Q, z, P[0] * P[1] * z % Q)
        kind = t % 5
        Qp = (0, 1) if kind == 0 else (P if kind == 1 else ((Q - P[0]) % Q, P[1]) if kind == 2 else ed_mul_aff(ED_B, rng.randrange(1, L_ORD)))
        em = _emu(rmn)
        for regs, v in ((rmn.X1, X), (rmn.Y1, Y), (rmn.Z1, Z), (rmn.T1, T)):
            em.setv(regs, _lazy(rng, v))
        for regs, v in ((rmn.QP, (Qp[1] + Qp[0]) % Q), (rmn.QM, (Qp[1] - Qp[0]) % Q), (rmn.QT, 2 * D_ED * Qp[0] * Qp[1] % Q)):
            em.setv(regs, _lazy(rng, v))
        em.run(En_.order)
        gx, gy, gz, gt = (em.getv(r_) % Q for r_ in (rmn.X1, rmn.Y1, rmn.Z1, rmn.T1))
        assert max(em.getv(r_) for r_ in (rmn.X1, rmn.Y1, rmn.Z1, rmn.T1)) < LIM
        zi = pow(gz, -1, Q)
        want = ed_add_aff(P, Qp)
        assert (gx * zi % Q, gy * zi % Q) == want and gt * zi % Q == want[0] * want[1] % Q, ("niels", t, kind)
    return out


# ---- the loop ---------------------------------------------------------------------------------------------------------------
def emit_loop():
    """Operands: %[tid] (VGPR), %[n] (SGPR), %[ptid] (VGPR) / %[np] (SGPR): the lane's table column and the number of columns (lanes that
    multiply the same point share one), %[tab] %[dig] %[res] (SGPR pairs).  tab: [17][np] cached entries of 128 B; dig: [51][n] records
    (bits 0-4 table index = |digit|, bit 5 negate); res: [n] (X, Y, Z, T) of 128 B, values below 2^255."""
    def go():
        rm = RegMap()
        L = []
        A = L.append
        lbl = lambda s: "%s_%%=" % s
        quad = G.quad
        inv = (-pow(Q, -1, 1 << 32)) & M32
        one = 1
        A("s_nop 1")
        A("s_mov_b32 %s, 0x%08x" % (S_INV, inv))
        for j in range(8):
            A("s_mov_b32 %s, 0x%08x" % (S_P[j], (Q >> (32 * j)) & M32))
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.TWOQ[j], ((2 * Q) >> (32 * j)) & M32))
        for t in rm.C:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
        A("v_lshlrev_b32_e32 %s, 7, %%[ptid]" % rm.tid128)          # table column; the result offset is formed at the end
        A("s_lshl_b32 %s, %%[n], 2" % S_N4)
        A("s_lshl_b32 %s, %%[np], 7" % S_N128)
        for j in range(8):                                            # accumulator = identity (0, 1, 1, 0)
            A("v_mov_b32_e32 %s, 0" % rm.X1[j])
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.Y1[j], (one >> (32 * j)) & M32))
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.Z1[j], (one >> (32 * j)) & M32))
            A("v_mov_b32_e32 %s, 0" % rm.T1[j])
        A("s_mov_b32 %s, 0" % S_STEP)
        EC.align_head(A)
        A(lbl("E_step") + ":")
        A("s_mul_i32 %s, %s, %s" % (S_TMP, S_STEP, S_N4))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid4))
        A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.off))
        A("s_cmp_eq_u32 %s, 0" % S_STEP)
        A("s_cbranch_scc1 " + lbl("E_nodbl"))
        A("s_mov_b32 %s, 5" % S_DBL)
        EC.align_head(A)
        A(lbl("E_dbl") + ":")
        Ed = Emitter(); Ed.schedule(seq_double(rm, False)); L.extend(Ed.lines)
        # T = E Hn only before the addition: the last of the five doublings
        A("s_cmp_lg_u32 %s, 1" % S_DBL)
        A("s_cbranch_scc1 " + lbl("E_dbl_not"))
        Et = Emitter(); Et.schedule(pmul(rm, rm.QM, rm.Cv, rm.T1)); L.extend(Et.lines)
        A(lbl("E_dbl_not") + ":")
        A("s_sub_u32 %s, %s, 1" % (S_DBL, S_DBL))
        A("s_cmp_lg_u32 %s, 0" % S_DBL)
        A("s_cbranch_scc1 " + lbl("E_dbl"))
        A(lbl("E_nodbl") + ":")
        A("s_waitcnt vmcnt(0)")
        A("v_and_b32_e32 %s, 31, %s" % (rm.tmp, rm.rec))
        A("v_mul_lo_u32 %s, %s, %s" % (rm.tmp, rm.tmp, S_N128))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, rm.tmp, rm.tid128))
        for k, regs in enumerate((rm.QP, rm.QM, rm.QT, rm.QZ)):
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[:4]), rm.off, 32 * k))
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[4:]), rm.off, 32 * k + 16))
        A("v_and_b32_e32 %s, 32, %s" % (rm.tmp, rm.rec))
        A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NEG, rm.tmp))
        A("s_waitcnt vmcnt(0)")
        # negative digit: -(x, y) has the cached form (Y-X, Y+X, -2dT, 2Z): swap the first two, 2dT -> 2q - 2dT (entries are canonical: < q)
        En = Emitter()
        seq = [i_subco(rm.A[0], rm.TWOQ[0], rm.QT[0], "vcc")] + [i_subb(rm.A[j], rm.TWOQ[j], rm.QT[j], "vcc") for j in range(1, 8)]
        seq += fold(rm, rm.A, None, rm.A, S_CY2)                      # 2q - 0 = 2q would leave the value range (points with T = 0): fold it back
        seq += EC.movs(rm.Bv, rm.QP)
        En.lastw[S_NEG] = -1
        seq += [i_cnd(rm.QP[j], rm.QP[j], rm.QM[j], S_NEG) for j in range(8)]
        seq += [i_cnd(rm.QM[j], rm.QM[j], rm.Bv[j], S_NEG) for j in range(8)]
        seq += [i_cnd(rm.QT[j], rm.QT[j], rm.A[j], S_NEG) for j in range(8)]
        En.schedule(seq); L.extend(En.lines)
        # 2q - 0 = 2q is not below 2^255 + 19: the identity entry (2dT = 0) is never negated (its record carries no sign)
        Ea = Emitter(); Ea.schedule(seq_add(rm, False)); L.extend(Ea.lines)
        # the very last addition also produces T (the result is stored in extended coordinates)
        A("s_cmp_lg_u32 %s, %d" % (S_STEP, N_WINDOWS - 1))
        A("s_cbranch_scc1 " + lbl("E_add_not"))
        # T3 = E H with E = QP, H = QM as seq_add leaves them
        Eat = Emitter(); Eat.schedule(pmul(rm, rm.QP, rm.QM, rm.T1)); L.extend(Eat.lines)
        A(lbl("E_add_not") + ":")
        A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
        A("s_cmp_lt_u32 %s, %d" % (S_STEP, N_WINDOWS))
        A("s_cbranch_scc1 " + lbl("E_step"))
        A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
        for k, regs in enumerate((rm.X1, rm.Y1, rm.T1, rm.Z1)):        # ark-ec order: x, y, t, z
            A("global_store_dwordx4 %s, %s, %%[res] offset:%d" % (rm.tid128, quad(regs[:4]), 32 * k))
            A("global_store_dwordx4 %s, %s, %%[res] offset:%d" % (rm.tid128, quad(regs[4:]), 32 * k + 16))
        A("s_waitcnt vmcnt(0)")
        st = dict(double=len(Ed.order), double_nops=Ed.nops, add=len(Ea.order), add_nops=Ea.nops, vgpr_end=rm.end)
        return L, rm, st
    return _with_globals(go)


def seq_add_niels(rm):
    """Accumulator += an AFFINE point in Niels form (y+x, y-x, 2dxy) held in QP, QM, QT (madd-2008-hwcd-3, Z2 = 1: D = 2 Z1 is a doubling, not a
    product): 7 products, T included.  Complete like seq_add; the identity's Niels form is (1, 1, 0).  In place."""
    X, Y, Z, T = rm.X1, rm.Y1, rm.Z1, rm.T1
    A, B, t = rm.A, rm.Bv, rm.Cv
    s = []
    s += sub_lz(rm, Y, X, A, t)
    s += add_lz(rm, Y, X, B, t)
    s += pmul(rm, A, rm.QM, A)
    s += pmul(rm, B, rm.QP, B)
    s += pmul(rm, T, rm.QT, rm.QT)               # C
    s += add_lz(rm, Z, Z, rm.QZ, t)              # D = 2 Z1
    E, H, F, Gv = rm.QP, rm.QM, X, Y
    s += sub_lz(rm, B, A, E, t)
    s += add_lz(rm, B, A, H, t)
    s += sub_lz(rm, rm.QZ, rm.QT, F, t)
    s += add_lz(rm, rm.QZ, rm.QT, Gv, t)
    s += pmul(rm, F, Gv, Z)
    s += pmul(rm, E, H, T)
    s += pmul(rm, E, F, X)
    s += pmul(rm, Gv, H, Y)
    return s


GEN_C = 11                           # signed 11-bit digits of the 253-bit scalars
GEN_WINDOWS = 23
GEN_ENTRIES = (1 << (GEN_C - 1)) + 2  # index 0 = the identity (a zero digit adds it: no predication), 1 .. 1025 = |digit| * 2^(11 w) * B (1025: the top window's carry)


def emit_gen_chain():
    """Fixed-base multiplication by the base point: 23 additions of tabulated affine multiples (plain-domain Niels entries of 96 B, table
    [23][1025] = 2.2 MiB, L2-resident), no doublings.  Operands: %[tid] (VGPR), %[n] (SGPR), %[dig] (SGPR pair: [23][n] records = table entry
    index | sign << 31), %[tab] %[res] (SGPR pairs; res: [n] (X, Y, T, Z) of 128 B, plain values below the loop's bound)."""
    def go():
        rm = RegMap()
        L = []
        A = L.append
        lbl = lambda s_: "%s_%%=" % s_
        quad = G.quad
        A("s_nop 1")
        for j in range(8):
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.TWOQ[j], ((2 * Q) >> (32 * j)) & M32))
        for t in rm.C:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
        A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
        A("s_lshl_b32 %s, %%[n], 2" % S_N4)
        for j in range(8):                                            # accumulator = identity (0, 1, 1, 0)
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
        for k, regs in enumerate((rm.QP, rm.QM, rm.QT)):
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[:4]), rm.off, 32 * k))
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[4:]), rm.off, 32 * k + 16))
        A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
        # the next window's record travels while this addition runs
        A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
        A("s_min_u32 %s, %s, %d" % (S_TMP, S_TMP, GEN_WINDOWS - 1))
        A("s_mul_i32 %s, %s, %s" % (S_TMP, S_TMP, S_N4))
        A("v_add_u32_e32 %s, %s, %s" % (rm.tmp, S_TMP, rm.tid4))
        A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.tmp))
        A("s_waitcnt vmcnt(1)")
        # negative digit: -(x, y) has the Niels form (y-x, y+x, -2dxy)
        En = Emitter()
        seq = [i_subco(rm.A[0], rm.TWOQ[0], rm.QT[0], "vcc")] + [i_subb(rm.A[j], rm.TWOQ[j], rm.QT[j], "vcc") for j in range(1, 8)]
        seq += fold(rm, rm.A, None, rm.A, S_CY2)
        seq += EC.movs(rm.Bv, rm.QP)
        En.lastw[S_NEG] = -1
        seq += [i_cnd(rm.QP[j], rm.QP[j], rm.QM[j], S_NEG) for j in range(8)]
        seq += [i_cnd(rm.QM[j], rm.QM[j], rm.Bv[j], S_NEG) for j in range(8)]
        seq += [i_cnd(rm.QT[j], rm.QT[j], rm.A[j], S_NEG) for j in range(8)]
        En.schedule(seq); L.extend(En.lines)
        Ea = Emitter(); Ea.schedule(seq_add_niels(rm)); L.extend(Ea.lines)
        A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
        A("s_cmp_lt_u32 %s, %d" % (S_STEP, GEN_WINDOWS))
        A("s_cbranch_scc1 " + lbl("G_step"))
        A("s_waitcnt vmcnt(0)")
        for k, regs in enumerate((rm.X1, rm.Y1, rm.T1, rm.Z1)):
            A("global_store_dwordx4 %s, %s, %%[res] offset:%d" % (rm.tid128, quad(regs[:4]), 32 * k))
            A("global_store_dwordx4 %s, %s, %%[res] offset:%d" % (rm.tid128, quad(regs[4:]), 32 * k + 16))
        A("s_waitcnt vmcnt(0)")
        mult = sum(1 for i in Ea.order if i.op in ("mad", "mul_lo"))
        return L, rm, dict(add=len(Ea.order), chain_mults=GEN_WINDOWS * mult, vgpr_end=rm.end)
    return _with_globals(go)


def cached_seq(rm, S_2D):
    """(A, Bv, Cv, QZ) = (Y1 + X1, Y1 - X1, 2d T1, 2 Z1) of the accumulator; QP / QM are scratch.  S_2D: SGPRs holding the plain limbs of 2d."""
    seq = add_lz(rm, rm.Y1, rm.X1, rm.A, rm.QP) + sub_lz(rm, rm.Y1, rm.X1, rm.Bv, rm.QM)
    seq += pmul(rm, rm.T1, S_2D, rm.Cv) + add_lz(rm, rm.Z1, rm.Z1, rm.QZ, rm.QP)
    return seq


def emit_table():
    """The window table of one scalar-mul as one asm stream: cached entries (Y+X, Y-X, 2dT, 2Z) of 0*P .. 16*P, in the same plain-product
    arithmetic as the loop (the entries only have to be SOME projective representative: the identity entry is (1, 1, 0, 2), the others carry
    whatever scale the chain P, 2P = dbl(P), kP = (k-1)P + P produces).  Operands: %[tid] (VGPR), %[poff] (VGPR: byte offset of this lane's
    point -- ark-ec order x, y, t, z, Montgomery-form limbs read as plain elements), %[n] (SGPR), %[pts] %[tab] (SGPR pairs)."""
    def go():
        rm = RegMap()
        L = []
        A = L.append
        lbl = lambda s_: "%s_%%=" % s_
        quad = G.quad
        S_2D = S_P                                                     # the q limbs are not needed without Montgomery reduction: 2d lives there
        d2 = 2 * D_ED % Q
        A("s_nop 1")
        for j in range(8):
            A("s_mov_b32 %s, 0x%08x" % (S_2D[j], (d2 >> (32 * j)) & M32))
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.TWOQ[j], ((2 * Q) >> (32 * j)) & M32))
        for t in rm.C:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("v_lshlrev_b32_e32 %s, 7, %%[tid]" % rm.tid128)
        A("s_lshl_b32 %s, %%[n], 7" % S_N128)
        for k, regs in enumerate((rm.X1, rm.Y1, rm.T1, rm.Z1)):
            A("global_load_dwordx4 %s, %%[poff], %%[pts] offset:%d" % (quad(regs[:4]), 32 * k))
            A("global_load_dwordx4 %s, %%[poff], %%[pts] offset:%d" % (quad(regs[4:]), 32 * k + 16))

        def store_entry(regs4, index):
            if isinstance(index, int):
                A("s_mul_i32 %s, %s, %d" % (S_TMP, S_N128, index))
            else:
                A("s_mul_i32 %s, %s, %s" % (S_TMP, S_N128, index))
            A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid128))
            for k, regs in enumerate(regs4):
                A("global_store_dwordx4 %s, %s, %%[tab] offset:%d" % (rm.off, quad(regs[:4]), 32 * k))
                A("global_store_dwordx4 %s, %s, %%[tab] offset:%d" % (rm.off, quad(regs[4:]), 32 * k + 16))
            A("s_nop 1")

        def cached(index):
            """cached form of the accumulator into A, Bv, Cv, QZ (QP / QM are scratch), stored as entry `index`"""
            E = Emitter()
            E.schedule(cached_seq(rm, S_2D))
            L.extend(E.lines)
            store_entry((rm.A, rm.Bv, rm.Cv, rm.QZ), index)
            return E

        # entry 0: the identity (0 : 1 : 1 : 0) -> (1, 1, 0, 2)
        for regs, v in ((rm.A, 1), (rm.Bv, 1), (rm.Cv, 0), (rm.QZ, 2)):
            for j in range(8):
                A("v_mov_b32_e32 %s, %d" % (regs[j], v if j == 0 else 0))
        store_entry((rm.A, rm.Bv, rm.Cv, rm.QZ), 0)
        A("s_waitcnt vmcnt(0)")
        Ec = cached(1)
        Ed = Emitter(); Ed.schedule(seq_double(rm, True)); L.extend(Ed.lines)
        cached(2)
        A("s_mov_b32 %s, 3" % S_STEP)
        A(lbl("T_next") + ":")
        A("s_waitcnt vmcnt(0)")                                          # entry 1 has landed (and the stores above have read their registers)
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_N128, rm.tid128))
        for k, regs in enumerate((rm.QP, rm.QM, rm.QT, rm.QZ)):
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[:4]), rm.off, 32 * k))
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs[4:]), rm.off, 32 * k + 16))
        A("s_waitcnt vmcnt(0)")
        Ea = Emitter(); Ea.schedule(seq_add(rm, True)); L.extend(Ea.lines)
        cached(S_STEP)
        A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
        A("s_cmp_le_u32 %s, 16" % S_STEP)
        A("s_cbranch_scc1 " + lbl("T_next"))
        A("s_waitcnt vmcnt(0)")
        mult = lambda E: sum(1 for i in E.order if i.op in ("mad", "mul_lo"))
        st = dict(table_mults=16 * mult(Ec) + mult(Ed) + 14 * mult(Ea), vgpr_end=rm.end)
        return L, rm, st
    return _with_globals(go)


def emit_header(path):
    selftest(trials=14)
    lines, rm, st = emit_loop()
    def mults(fn):
        def go():
            rm_ = RegMap()
            return sum(1 for i in fn(rm_) if i.op in ("mad", "mul_lo"))
        return _with_globals(go)
    m_mul = mults(lambda r_: pmul(r_, r_.X1, r_.Y1, r_.Z1))
    loop_m = (N_WINDOWS - 1) * 5 * mults(lambda r_: seq_double(r_, False)) + (N_WINDOWS - 1) * m_mul + N_WINDOWS * mults(lambda r_: seq_add(r_, False)) + m_mul
    out = ["// GENERATED by tools/gen_ed_asm.py -- do not edit.  The Curve25519 (twisted Edwards) window loop as one hand-scheduled gfx950 stream;",
           "// see the generator for the value range (< 2^255 + 19), the complete addition law (no exceptional lanes) and the emulator check.",
           "// double: %d instructions (%d wait states), add: %d (%d), VGPRs v%d..v%d; %d multiplier instructions per scalar-mul in the loop." %
           (st["double"], st["double_nops"], st["add"], st["add_nops"], rm.first, rm.end - 1, loop_m),
           "#pragma once", "#define ED_ASM_WINDOWS %d" % N_WINDOWS, "#define ED_ASM_TABLE %d" % N_TABLE, "#define ED_ASM_MULT_INSTRS_LOOP %d" % loop_m,
           "__device__ __forceinline__ void ed_smul_loop_asm(u32 tid, u32 n, u32 ptid, u32 np, const u64* tab, const u32* dig, u64* res) {", "    asm volatile(",
           G.c_string(lines), "        :", '        : [tid] "v"(tid), [n] "s"(n), [ptid] "v"(ptid), [np] "s"(np), [tab] "s"(tab), [dig] "s"(dig), [res] "s"(res)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(rm.first, rm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    tlines, trm, tst = emit_table()
    out += ["// the window table (cached entries of 0*P .. 16*P) in the same arithmetic: %d asm lines, %d multiplier instructions per scalar-mul" % (len(tlines), tst["table_mults"]),
            "#define ED_ASM_MULT_INSTRS_TABLE %d" % tst["table_mults"],
            "__device__ __forceinline__ void ed_smul_table_asm(u32 tid, u32 poff, u32 n, const u64* pts, u64* tab) {", "    asm volatile(",
            G.c_string(tlines), "        :", '        : [tid] "v"(tid), [poff] "v"(poff), [n] "s"(n), [pts] "s"(pts), [tab] "s"(tab)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(trm.first, trm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    glines, grm, gst = emit_gen_chain()
    out += ["// fixed-base multiplication by the base point: %d additions of tabulated affine multiples (plain Niels entries), %d asm lines, %d multiplier instructions" %
            (GEN_WINDOWS, len(glines), gst["chain_mults"]),
            "#define ED_GEN_ASM_C %d" % GEN_C, "#define ED_GEN_ASM_WINDOWS %d" % GEN_WINDOWS, "#define ED_GEN_ASM_ENTRIES %d" % GEN_ENTRIES,
            "__device__ __forceinline__ void ed_gen_chain_asm(u32 tid, u32 n, const u32* dig, const u64* tab, u64* res) {", "    asm volatile(",
            G.c_string(glines), "        :", '        : [tid] "v"(tid), [n] "s"(n), [dig] "s"(dig), [tab] "s"(tab), [res] "s"(res)']
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(grm.first, grm.end)]
    out += ["        : " + ", ".join(clob) + ");", "}"]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return st, len(lines), loop_m


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "ed_asm_kernels.inc"))
    a = ap.parse_args()
    if a.selftest:
        r = selftest(trials=120)
        print("ok:", {k: (len(v[0].order), v[0].nops, len(v[1].order), v[1].nops) for k, v in r.items()})
        sys.exit(0)
    st, n, loop_m = emit_header(a.o)
    print("ed loop: %d asm lines; %s; %d multiplier instructions per scalar-mul" % (n, st, loop_m))
