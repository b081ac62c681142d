#!/usr/bin/env python3
"""Generator for the hand-scheduled BN254 G1 scalar-multiplication loop (ark-mpc_amd/csrc/ec_asm_kernels.inc).

Why: the compiled window loop (k_g1_scalar_mul) reaches 0.41 of the v_mad_u64_u32 peak while a bare chain of the
hand-scheduled Montgomery block reaches 0.61 from two waves per SIMD upwards (probes/mulrate.hip).  The gap is everything around the
multiplier: by-value calls of __noinline__ point functions (moves, scratch spills at 248 VGPRs), general Jacobian additions,
issue stalls.  Here the whole loop is one instruction stream over fixed registers:

  * the window table is EFFECTIVE-AFFINE: the prep kernel rescales the 16 multiples of P to one common Z, so on the isomorphic
    curve y^2 = x^3 + 3 Zc^6 they are affine points (x_k, y_k) and every window addition is a MIXED addition (madd-2007-bl,
    11 multiplications instead of 16; neither it nor dbl-2009-l uses the curve constant when a = 0); Z is multiplied back by Zc
    in the epilogue kernel;
  * the accumulator never is the identity: it starts at a fixed point R0 (rescaled to the isomorphic curve by the prep kernel) and
    the known multiple 2^130 R0 is subtracted by the last step -- so there is no infinity bookkeeping in the loop;
  * every step is the same code: load one digit record, load one affine table entry, (multiply x by beta for the phi half),
    conditionally negate y, mixed-add, keep the old accumulator where the digit is zero;
  * the exceptional case of the group law (H = 0: equal or opposite points) only raises a per-lane flag; the epilogue kernel
    recomputes flagged lanes on the compiled path, so results are exact for EVERY input.

Coordinates live in the lazy range [0, 2q) (4q < 2^256), as in arkmpc_curve.hip.  The two bodies (double, mixed add) are
built from the multiplier rows of gen_asm_kernels.py, scheduled by its hazard-aware list scheduler and executed by its
single-lane emulator against the affine group law in Python integers before they are emitted (--selftest,
tests/test_asm_generator.py).

Reference semantics: CurvePoint * Scalar (online-phase/src/algebra/curve/curve.rs:403-409), PointShare * Scalar
(curve/share.rs:108-114).
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_asm_kernels as G
from gen_asm_kernels import Ins, Emitter, Emu, M32, R, i_mov, i_mad, i_addco, i_addc, i_subco, i_subb, i_cnd, regs_of

Q = dict(G.FIELDS)["BN254_FQ"]
RORD = dict(G.FIELDS)["BN254_FR"]
TWOQ = 2 * Q
assert 4 * Q < R

# ---- SGPR map (all clobbered by the asm statement) ----------------------------------------------------------------------
S_JUNK, S_CY2, S_INV = "s[16:17]", "s[18:19]", "s20"
S_P = ["s%d" % (21 + i) for i in range(8)]            # q limbs (multiplier reduction rows, zero test)
S_BETA = ["s%d" % (29 + i) for i in range(8)]         # beta in Montgomery form (phi half: x -> beta x)
S_STEP, S_N4, S_N64, S_TMP, S_DBL = "s37", "s38", "s39", "s40", "s41"
S_NZ, S_NEG, S_EXC, S_M1, S_M2 = "s[42:43]", "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]"
S_CY3, S_CY4 = "s[52:53]", "s[54:55]"                 # extra carry registers: two lazy add / sub operations can be in flight at once
CLOBBER_SGPRS = ["s%d" % i for i in range(16, 56)]
_CARRY_POOL = [("vcc", S_CY2), (S_CY3, S_CY4)]
_carry_turn = [0]


def i_shl(d, a, sh): return Ins("v_lshlrev_b32_e32 %s, %d, %s" % (d, sh, a), "shl", (d, a, sh), rd=regs_of(a), wr=[d])


def montsqr(rm, a, out):
    """out = a^2 / R in [0, 2q) for a in [0, 2q); out may alias a.  Squaring rows: with d = 2a (8 limbs: a < 2q < 2^255), row r
    multiplies a_r only into positions j >= r -- a_r a_r at j = r, d_j a_r above it; the products with j < r were added, doubled, by
    the earlier rows.  Position r (the limb the reduction row of step r zeroes) is complete at that point: every pair (i, j) with
    i + j = r has min(i, j) <= r / 2.  36 instead of 64 products, and as many additions fewer; the accumulator bound
    (rows add at most (2^32 - 1) * 2a) stays below 2^288 for this modulus (checked below)."""
    t = 0
    for _ in range(64):                                   # exact worst case of the 9-limb accumulator
        before = t + M32 * (2 * (TWOQ - 1))
        after = (before + M32 * Q) >> 32
        if after == t:
            break
        t = after
    assert before < (1 << 288)
    T, Tz, q, m, d = rm.T, rm.Tz, rm.q, rm.m, rm.D2
    seq = [i_shl(d[0], a[0], 1)] + [i_alignbit(d[j], a[j], a[j - 1], 31) for j in range(1, 8)]
    row = 0
    for r in range(8):
        cy = "vcc" if row % 2 == 0 else S_CY2
        row += 1
        first = r == 0
        # d_{r+1} carries the top bit of a_r in its low bit (it belongs to 2 a_r, which this row does not add): mask it off
        if r < 7:
            seq += [G.i_and(m, -2 & M32, d[r + 1])]
        seq += [i_mad(q[j], a[r] if j == r else (m if j == r + 1 else d[j]), a[r], 0 if first else Tz[j]) for j in range(r, 8)]
        seq += [i_mov(T[r], q[r][0])]
        if r < 7:
            seq += [i_addco(T[r + 1], q[r + 1][0], q[r][1], cy)]
            seq += [i_addc(T[j], q[j][0], q[j - 1][1], cy) for j in range(r + 2, 8)]
            seq += [i_addc(T[8], 0 if first else T[8], q[7][1], cy)]
        else:
            seq += [i_addco(T[8], T[8], q[7][1], cy)]
        cy = "vcc" if row % 2 == 0 else S_CY2
        row += 1
        seq += [G.i_mul_lo(m, T[0], S_INV)]
        seq += [i_mad(q[j], m, S_P[j], Tz[j]) for j in range(8)]
        D = out if r == 7 else T
        seq += [i_addco(D[0], q[1][0], q[0][1], cy)]
        seq += [i_addc(D[j - 1], q[j][0], q[j - 1][1], cy) for j in range(2, 8)]
        seq += [i_addc(D[7], T[8], q[7][1], cy)]
        if D is T:
            seq += [i_addc(T[8], 0, Tz[8][1], cy)]
    return seq


def _carries():
    """Alternate the carry-register pair from one lazy operation to the next, so that the list scheduler can overlap two
    independent operations (each is two dependent carry chains; with one pair they would serialise on the registers)."""
    _carry_turn[0] ^= 1
    return _CARRY_POOL[_carry_turn[0]]

N_STEPS = 55            # 27 windows x 2 halves + the final subtraction of 2^130 R0
N_TABLE = 18            # 16 multiples, R0', -(2^130 R0)'
BETA = None             # filled in main() from glv_consts (cube root of unity in Fq, Montgomery form)


class RegMap:
    """Fixed VGPR allocation of the loop (v8 upwards; v0-v7 stay with the compiler for the kernel's own few values)."""

    def __init__(self, first=8, table_kernel=False):
        rg = G.Regs(first)
        self.TWOQ = rg.vec(8)
        self.X1, self.Y1, self.Z1 = rg.vec(8, 4), rg.vec(8, 4), rg.vec(8, 4)      # accumulator (Jacobian on the isomorphic curve)
        self.X2, self.Y2 = rg.vec(8, 4), rg.vec(8, 4)                              # table entry (affine)
        self.SX, self.SY, self.SZ = rg.vec(8, 4), rg.vec(8, 4), rg.vec(8, 4)       # accumulator saved across the addition
        self.T0, self.T1, self.T2 = rg.vec(8, 4), rg.vec(8, 4), rg.vec(8, 4)
        self.D2 = rg.vec(8, 4)                                                     # 2a for the squaring rows
        self.Tz = [rg.pair() for _ in range(9)]
        self.T = [t[0] for t in self.Tz]
        self.q = [rg.pair() for _ in range(8)]
        self.m = rg.one()
        self.rec, self.off, self.tid4, self.tid64, self.tid96, self.tmp, self.flag = (rg.one() for _ in range(7))
        if table_kernel:
            self.QV = rg.vec(8, 4)                                                 # q limbs in VGPRs (carry ops cannot take SGPR operands)
            self.off2, self.tid32 = rg.one(), rg.one()
        self.first, self.end = first, rg.next


def montmul(rm, a, b, out):
    """out = a * b / R in [0, 2q) for a, b in [0, 2q); out may alias an operand.  b may be a list of SGPR names."""
    seq, _ = G.montmul_sum_seq(Q, [(a, b)], rm.T, rm.Tz, rm.q, rm.m, S_P, 0, final_out=out)
    return seq


def add_lz(rm, a, b, out, tmp):          # (a + b) mod 2q
    c1, c2 = _carries()
    return G.fe_add_seq(rm.TWOQ, a, b, out, tmp, c1, c2)


def sub_lz(rm, a, b, out, tmp):          # a - b (+ 2q on borrow); out may alias a or b, tmp may not
    c1, c2 = _carries()
    seq = [i_subco(out[0], a[0], b[0], c1)] + [i_subb(out[j], a[j], b[j], c1) for j in range(1, 8)]
    seq += [i_cnd(tmp[j], 0, rm.TWOQ[j], c1) for j in range(8)]
    seq += [i_addco(out[0], out[0], tmp[0], c2)] + [i_addc(out[j], out[j], tmp[j], c2) for j in range(1, 8)]
    return seq


def i_bfe_i(d, a, off, width): return Ins("v_bfe_i32 %s, %s, %d, %d" % (d, a, off, width), "bfe_i", (d, a, off, width), rd=regs_of(a), wr=[d])
def i_alignbit(d, hi, lo, sh): return Ins("v_alignbit_b32 %s, %s, %s, %d" % (d, hi, lo, sh), "alignbit", (d, hi, lo, sh), rd=regs_of(hi, lo), wr=[d])
def i_lshr(d, a, sh): return Ins("v_lshrrev_b32_e32 %s, %d, %s" % (d, sh, a), "lshr", (d, a, sh), rd=regs_of(a), wr=[d])


def half_lz(rm, a, out, tmp):
    """a / 2 mod q for a in [0, 2q): add q when a is odd (a + q < 3q < 2^256), shift right.  Result < 1.5 q.  One carry chain."""
    c1, _ = _carries()
    seq = [i_bfe_i(tmp[0], a[0], 0, 1)]                                          # all ones when a is odd
    seq += [G.i_and(tmp[j], S_P[j], tmp[0]) for j in range(7, -1, -1)]           # q & mask (tmp[0] last: it is the mask)
    seq += [i_addco(tmp[0], a[0], tmp[0], c1)] + [i_addc(tmp[j], a[j], tmp[j], c1) for j in range(1, 8)]
    seq += [i_alignbit(out[j], tmp[j + 1], tmp[j], 1) for j in range(7)] + [i_lshr(out[7], tmp[7], 1)]
    return seq


def dbl_lz(rm, a, out, tmp):
    return add_lz(rm, a, a, out, tmp)


def movs(dst, src_):
    return [i_mov(d, s) for d, s in zip(dst, src_)]


# ---- the two bodies -------------------------------------------------------------------------------------------------------
def seq_double(rm):
    """dbl-2009-l with a = 0, in place on (X1, Y1, Z1); scratch: X2, Y2, SX, SY, SZ, T0..T2.  Arranged around Y' = 2 Y so that the
    powers of two come out of the multiplications: B4 = Y'^2 = 4 B, Z3 = Y' Z, D = X B4 = 4 X B, C16 = B4^2 = 16 C and 8 C = C16 / 2
    (one halving instead of three doublings).  7 multiplier blocks (4 of them squarings), 8 lazy operations."""
    X, Y, Z = rm.X1, rm.Y1, rm.Z1
    A, B4, C16, D, E, F = rm.T0, rm.T1, rm.X2, rm.Y2, rm.SX, rm.SY
    t, u = rm.T2, rm.SZ
    s = []
    s += montsqr(rm, X, A)                       # A = X^2
    s += dbl_lz(rm, Y, Y, t)                     # Y' = 2 Y
    s += montsqr(rm, Y, B4)                      # 4 B
    s += montmul(rm, Y, Z, Z)                    # Z3 = 2 Y Z
    s += montmul(rm, X, B4, D)                   # D = 4 X B
    s += dbl_lz(rm, A, E, t) + add_lz(rm, E, A, E, u)         # E = 3 A
    s += montsqr(rm, B4, C16)                   # 16 C
    s += montsqr(rm, E, F)                       # F = E^2
    s += dbl_lz(rm, D, u, t)                     # 2 D
    s += sub_lz(rm, F, u, X, t)                  # X3 = F - 2 D
    s += half_lz(rm, C16, C16, u)                # 8 C
    s += sub_lz(rm, D, X, D, t)                  # D - X3
    s += montmul(rm, E, D, Y)                    # E (D - X3)
    s += sub_lz(rm, Y, C16, Y, t)                # Y3
    return s


def i_or(d, a, b): return Ins("v_or_b32_e32 %s, %s, %s" % (d, a, b), "or", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_xor(d, a, b): return Ins("v_xor_b32_e32 %s, %s, %s" % (d, a, b), "xor", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_cmpz(mask, a): return Ins("v_cmp_eq_u32_e64 %s, 0, %s" % (mask, a), "cmpz", (mask, a), swr=(mask,), rd=regs_of(a))


def seq_madd(rm):
    """madd-2007-bl: (X1, Y1, Z1) += affine (X2, Y2), in place; scratch T0..T2.  With H2 = 2 H: I = H2^2 (= 4 HH), Z3 = Z1 H2
    (= 2 Z1 H, one multiplication instead of (Z1+H)^2 - Z1Z1 - HH).  11 multiplier blocks (3 squarings), 10 lazy operations.
    Leaves the two zero tests of H (== 0, == q: the lazy range holds both representatives of 0 mod q) as masks in S_M1 / S_M2."""
    X1, Y1, Z1, X2, Y2 = rm.X1, rm.Y1, rm.Z1, rm.X2, rm.Y2
    ZZ, t, u = rm.T0, rm.T1, rm.T2
    s = []
    s += montsqr(rm, Z1, ZZ)                     # Z1Z1
    s += montmul(rm, X2, ZZ, X2)                 # U2
    s += montmul(rm, Y2, Z1, Y2)                 # Y2 Z1
    H = X2
    s += sub_lz(rm, X2, X1, H, t)                # H = U2 - X1
    # zero tests of H for the exceptional-case flag (values are < 2q: H = 0 mod q  <=>  H in {0, q})
    s += [i_or(u[0], H[0], H[1])] + [i_or(u[0], u[0], H[j]) for j in range(2, 8)] + [i_cmpz(S_M1, u[0])]
    s += [i_xor(u[j], S_P[j], H[j]) for j in range(8)]
    s += [i_or(u[0], u[0], u[j]) for j in range(1, 8)] + [i_cmpz(S_M2, u[0])]
    s += montmul(rm, Y2, ZZ, Y2)                 # S2
    H2 = ZZ
    s += dbl_lz(rm, H, H2, t)                    # H2 = 2 H        (Z1Z1 is dead)
    s += montmul(rm, Z1, H2, Z1)                 # Z3 = 2 Z1 H
    I = H2
    s += montsqr(rm, H2, I)                      # I = 4 HH
    rr = Y2
    s += sub_lz(rm, Y2, Y1, rr, t) + dbl_lz(rm, rr, rr, u)       # r = 2 (S2 - Y1)
    J, V = H, X1
    s += montmul(rm, X1, I, V)                   # V = X1 I      (X1 dead from here)
    s += montmul(rm, H, I, J)                    # J = H I
    s += montsqr(rm, rr, I)                      # r^2  (I dead)
    s += montmul(rm, Y1, J, Y1)                  # Y1 J
    s += sub_lz(rm, I, J, I, t)                  # r^2 - J
    s += dbl_lz(rm, V, u, t)                     # 2 V
    s += sub_lz(rm, I, u, I, t)                  # X3 = r^2 - J - 2 V     (in T0)
    s += sub_lz(rm, V, I, V, u)                  # V - X3
    s += dbl_lz(rm, Y1, Y1, t)                   # 2 Y1 J
    s += montmul(rm, rr, V, V)                   # r (V - X3)
    s += sub_lz(rm, V, Y1, Y1, t)                # Y3
    s += movs(X1, I)                             # X3 into the accumulator registers
    return s


# ---- emulator extensions ----------------------------------------------------------------------------------------------------
class EcEmu(Emu):
    def run(self, order):
        for ins in order:
            op, a = ins.op, ins.args
            if op == "or":
                self.v[a[0]] = self.rd(a[1]) | self.rd(a[2])
            elif op == "xor":
                self.v[a[0]] = self.rd(a[1]) ^ self.rd(a[2])
            elif op == "cmpz":
                self.c[a[0]] = 1 if self.rd(a[1]) == 0 else 0
            elif op == "bfe_i":
                self.v[a[0]] = M32 if (self.rd(a[1]) >> a[2]) & 1 else 0          # width 1 only
            elif op == "alignbit":
                self.v[a[0]] = (((self.rd(a[1]) << 32) | self.rd(a[2])) >> a[3]) & M32
            elif op == "lshr":
                self.v[a[0]] = self.rd(a[1]) >> a[2]
            elif op == "shl":
                self.v[a[0]] = (self.rd(a[1]) << a[2]) & M32
            else:
                Emu.run(self, [ins])


def _with_globals(fn):
    saved = (G.JUNK, G.S_INV, G.CY2)
    G.JUNK, G.S_INV, G.CY2 = S_JUNK, S_INV, S_CY2
    try:
        return fn()
    finally:
        G.JUNK, G.S_INV, G.CY2 = saved


def build_body(which, sched=True):
    def go():
        rm = RegMap()
        E = Emitter()
        seq = seq_double(rm) if which == "double" else seq_madd(rm)
        (E.schedule if sched else E.emit_all)(seq)
        return E, rm
    return _with_globals(go)


# ---- Python model of the group (affine, integers) -------------------------------------------------------------------------
def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % Q == 0: return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, Q) % Q
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, Q) % Q
    x = (lam * lam - a[0] - b[0]) % Q
    return (x, (lam * (a[0] - x) - a[1]) % Q)


def g1_mul(a, k):
    r = None
    while k:
        if k & 1: r = g1_add(r, a)
        a = g1_add(a, a); k >>= 1
    return r


GEN = (1, 2)
mont = lambda v: v * R % Q
unmont = lambda v: v * pow(R, -1, Q) % Q


def _emu_for(rm):
    em = EcEmu()
    em.setv(rm.TWOQ, TWOQ)
    for t in rm.Tz:
        em.v[t[1]] = 0
    for j in range(8):
        em.s[S_P[j]] = (Q >> (32 * j)) & M32
    em.s[S_INV] = (-pow(Q, -1, 1 << 32)) & M32
    return em


def _lazy(rng, v):
    """a representative of v (mod q) in [0, 2q)"""
    return v + Q if rng.random() < 0.5 else v


def _affine(X, Y, Z):
    X, Y, Z = unmont(X % Q), unmont(Y % Q), unmont(Z % Q)
    if Z == 0: return None
    zi = pow(Z, -1, Q)
    return (X * zi * zi % Q, Y * zi * zi * zi % Q)


def selftest(trials=40, seed=7):
    rng = random.Random(seed)
    Ed, rm = build_body("double")
    Ea, rm2 = build_body("madd")
    for t in range(trials):
        P = g1_mul(GEN, rng.randrange(1, RORD))
        z = rng.randrange(1, Q)
        Xj, Yj = P[0] * z * z % Q, P[1] * z * z * z % Q
        em = _emu_for(rm)
        em.setv(rm.X1, _lazy(rng, mont(Xj))); em.setv(rm.Y1, _lazy(rng, mont(Yj))); em.setv(rm.Z1, _lazy(rng, mont(z)))
        em.run(Ed.order)
        X3, Y3, Z3 = em.getv(rm.X1), em.getv(rm.Y1), em.getv(rm.Z1)
        assert max(X3, Y3, Z3) < TWOQ, "double: lazy range"
        assert _affine(X3, Y3, Z3) == g1_add(P, P), ("double", t)
        # mixed addition, incl. the exceptional inputs (same point / opposite point) which must raise the H = 0 masks
        kind = t % 8
        Qp = P if kind == 6 else ((P[0], Q - P[1]) if kind == 7 else g1_mul(GEN, rng.randrange(1, RORD)))
        em = _emu_for(rm2)
        em.setv(rm2.X1, _lazy(rng, mont(Xj))); em.setv(rm2.Y1, _lazy(rng, mont(Yj))); em.setv(rm2.Z1, _lazy(rng, mont(z)))
        em.setv(rm2.X2, _lazy(rng, mont(Qp[0]))); em.setv(rm2.Y2, _lazy(rng, mont(Qp[1])))
        em.run(Ea.order)
        exc = em.c[S_M1] | em.c[S_M2]
        if kind >= 6:
            assert exc == 1, ("madd must flag H = 0", t)
        else:
            X3, Y3, Z3 = em.getv(rm2.X1), em.getv(rm2.Y1), em.getv(rm2.Z1)
            assert exc == 0 and max(X3, Y3, Z3) < TWOQ
            assert _affine(X3, Y3, Z3) == g1_add(P, Qp), ("madd", t)
    return Ed, Ea


# ---- the loop ---------------------------------------------------------------------------------------------------------------
def raw(E, text):
    E.lines.append(text)


def quad(regs4):
    return G.quad(regs4)


def align_head(A):
    """Placement of a loop head: EC_ALIGN = "<log2 alignment>:<extra 4-byte nops>" (measured per setting on MI355X; hand-scheduled streams are
    sensitive to their fetch phase -- MI355X_MICROARCH.md, code-placement note)."""
    spec = os.environ.get("EC_ALIGN", LOOP_ALIGN)
    if not spec:
        return
    p2, nops = (int(x) for x in spec.split(":"))
    A(".p2align %d" % p2)
    for _ in range(nops):
        A("s_nop 0")


LOOP_ALIGN = "6:2"        # measured on MI355X (config 4): "6:0" 8.15 ms, none 7.73, "6:1" 7.65, "6:2" 7.61, "6:3" 7.65 (+-1 % run to run)


def emit_loop():
    """Returns (lines, regmap, stats): the whole step loop as assembler text.
    Operands: %[tid] (VGPR, lane's index in the launch), %[n] (SGPR, lanes in the launch), %[tab] %[dig] %[res] %[exc] (SGPR pairs);
    %[ptid] (VGPR) / %[np] (SGPR): the lane's column in the window table and the number of columns -- lanes that multiply the SAME point
    (ScalarShare x point: two lanes per point) share one table column."""
    def go():
        rm = RegMap()
        L = []
        A = L.append
        lbl = lambda s: "%s_%%=" % s                      # unique per expansion of the asm statement
        inv = (-pow(Q, -1, 1 << 32)) & M32
        A("s_nop 1")
        A("s_mov_b32 %s, 0x%08x" % (S_INV, inv))
        for j in range(8):
            A("s_mov_b32 %s, 0x%08x" % (S_P[j], (Q >> (32 * j)) & M32))
            A("s_mov_b32 %s, 0x%08x" % (S_BETA[j], (BETA >> (32 * j)) & M32))
        for j in range(8):
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.TWOQ[j], (TWOQ >> (32 * j)) & M32))
        for t in rm.Tz:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("v_lshlrev_b32_e32 %s, 2, %%[tid]" % rm.tid4)
        A("v_lshlrev_b32_e32 %s, 6, %%[ptid]" % rm.tid64)
        A("v_mul_u32_u24_e32 %s, 96, %%[tid]" % rm.tid96)
        A("s_lshl_b32 %s, %%[n], 2" % S_N4)
        A("s_lshl_b32 %s, %%[np], 6" % S_N64)
        A("s_mov_b64 %s, 0" % S_EXC)
        # accumulator = table entry 16 (R0 on the isomorphic curve), Z = 1 (Montgomery form)
        A("s_mul_i32 %s, %s, 16" % (S_TMP, S_N64))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid64))
        for k, regs in enumerate((rm.X1[:4], rm.X1[4:], rm.Y1[:4], rm.Y1[4:])):
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs), rm.off, 16 * k))
        one = R % Q
        for j in range(8):
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.Z1[j], (one >> (32 * j)) & M32))
        A("s_mov_b32 %s, 0" % S_STEP)
        align_head(A)
        A(lbl("L_step") + ":")
        # digit record of this step: bits 0-4 table index, bit 5 negate, bit 6 digit non-zero
        A("s_mul_i32 %s, %s, %s" % (S_TMP, S_STEP, S_N4))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, S_TMP, rm.tid4))
        A("global_load_dword %s, %s, %%[dig]" % (rm.rec, rm.off))
        # five doublings before the first half of every window but the top one (steps 2, 4, ..., 52)
        A("s_and_b32 %s, %s, 1" % (S_TMP, S_STEP))
        A("s_cmp_eq_u32 %s, 1" % S_TMP)
        A("s_cbranch_scc1 " + lbl("L_nodbl"))
        A("s_cmp_eq_u32 %s, 0" % S_STEP)
        A("s_cbranch_scc1 " + lbl("L_nodbl"))
        A("s_cmp_eq_u32 %s, %d" % (S_STEP, N_STEPS - 1))
        A("s_cbranch_scc1 " + lbl("L_nodbl"))
        A("s_mov_b32 %s, 5" % S_DBL)
        align_head(A)
        A(lbl("L_dbl") + ":")
        Ed = Emitter()
        Ed.schedule(seq_double(rm))
        L.extend(Ed.lines)
        A("s_sub_u32 %s, %s, 1" % (S_DBL, S_DBL))
        A("s_cmp_lg_u32 %s, 0" % S_DBL)
        A("s_cbranch_scc1 " + lbl("L_dbl"))
        A(lbl("L_nodbl") + ":")
        A("s_waitcnt vmcnt(0)")
        # table entry: offset = index * n * 64 + tid * 64
        A("v_and_b32_e32 %s, 31, %s" % (rm.tmp, rm.rec))
        A("v_mul_lo_u32 %s, %s, %s" % (rm.tmp, rm.tmp, S_N64))
        A("v_add_u32_e32 %s, %s, %s" % (rm.off, rm.tmp, rm.tid64))
        for k, regs in enumerate((rm.X2[:4], rm.X2[4:], rm.Y2[:4], rm.Y2[4:])):
            A("global_load_dwordx4 %s, %s, %%[tab] offset:%d" % (quad(regs), rm.off, 16 * k))
        # masks from the record; the accumulator is saved while the loads fly
        A("v_and_b32_e32 %s, 32, %s" % (rm.tmp, rm.rec))
        A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NEG, rm.tmp))
        A("v_and_b32_e32 %s, 64, %s" % (rm.tmp, rm.rec))
        A("v_cmp_ne_u32_e64 %s, 0, %s" % (S_NZ, rm.tmp))
        for d, s_ in zip(rm.SX + rm.SY + rm.SZ, rm.X1 + rm.Y1 + rm.Z1):
            A("v_mov_b32_e32 %s, %s" % (d, s_))
        A("s_waitcnt vmcnt(0)")
        # phi half (odd steps): x -> beta x
        A("s_and_b32 %s, %s, 1" % (S_TMP, S_STEP))
        A("s_cmp_eq_u32 %s, 0" % S_TMP)
        A("s_cbranch_scc1 " + lbl("L_nobeta"))
        Eb = Emitter()
        Eb.schedule(montmul(rm, rm.X2, S_BETA, rm.X2))
        L.extend(Eb.lines)
        A(lbl("L_nobeta") + ":")
        # y -> 2q - y where the record says so (table entries are canonical and y != 0 on this curve, so 2q - y stays in range)
        En = Emitter()
        seq = [i_subco(rm.T2[0], rm.TWOQ[0], rm.Y2[0], "vcc")] + [i_subb(rm.T2[j], rm.TWOQ[j], rm.Y2[j], "vcc") for j in range(1, 8)]
        En.lastw[S_NEG] = -1                                     # written by the v_cmp above: keep the H1 distance
        seq += [i_cnd(rm.Y2[j], rm.Y2[j], rm.T2[j], S_NEG) for j in range(8)]
        En.schedule(seq)
        L.extend(En.lines)
        Ea = Emitter()
        Ea.schedule(seq_madd(rm))
        L.extend(Ea.lines)
        # exceptional lanes: H = 0 (mod q) on a lane whose digit is non-zero
        A("s_nop 1")
        A("s_or_b64 %s, %s, %s" % (S_M1, S_M1, S_M2))
        A("s_and_b64 %s, %s, %s" % (S_M1, S_M1, S_NZ))
        A("s_or_b64 %s, %s, %s" % (S_EXC, S_EXC, S_M1))
        # digit zero: keep the saved accumulator
        for d, s_ in zip(rm.X1 + rm.Y1 + rm.Z1, rm.SX + rm.SY + rm.SZ):
            A("v_cndmask_b32_e64 %s, %s, %s, %s" % (d, s_, d, S_NZ))
        A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
        A("s_cmp_lt_u32 %s, %d" % (S_STEP, N_STEPS))
        A("s_cbranch_scc1 " + lbl("L_step"))
        # result (lazy, on the isomorphic curve) and the exceptional flag
        for k, regs in enumerate((rm.X1[:4], rm.X1[4:], rm.Y1[:4], rm.Y1[4:], rm.Z1[:4], rm.Z1[4:])):
            A("global_store_dwordx4 %s, %s, %%[res] offset:%d" % (rm.tid96, quad(regs), 16 * k))
        A("v_cndmask_b32_e64 %s, 0, 1, %s" % (rm.flag, S_EXC))
        A("global_store_dword %s, %s, %%[exc]" % (rm.tid4, rm.flag))
        A("s_waitcnt vmcnt(0)")
        stats = dict(double=len(Ed.order), double_nops=Ed.nops, madd=len(Ea.order), madd_nops=Ea.nops, beta=len(Eb.order), vgpr_end=rm.end)
        return L, rm, stats
    return _with_globals(go)


def emit_msm_acc():
    """Bucket accumulation of the variable-base MSM (arkmpc_msm.inc, step M3) on the mixed-addition body of the window loop: a lane folds
    the `len` members of its task -- vals[first .. first + len): point index | sign << 31 -- into one Jacobian sum.  The accumulator starts
    as the first member (Z = 1), so it is never the identity; H = 0 (a repeated point, or P + (-P)) only raises the lane's flag and the
    compiled kernel redoes that task.  The next member's index is fetched while the current one is added, and its coordinates are
    requested as soon as the addition has consumed the registers.
    Operands: %[lo4] (VGPR: byte offset of the first member in vals), %[len] (VGPR, >= 1), %[maxlen] (SGPR: longest task of the wave),
    %[vals] %[aff] %[exc] (SGPR pairs), %[dst] (VGPR pair: where this lane's sum goes), %[t4] (VGPR: 4 * task index)."""
    def go():
        rm = RegMap()
        rec2, last = rm.tid96, rm.tid64
        L = []
        A = L.append
        lbl = lambda s: "%s_%%=" % s
        inv = (-pow(Q, -1, 1 << 32)) & M32
        A("s_nop 1")
        A("s_mov_b32 %s, 0x%08x" % (S_INV, inv))
        for j in range(8):
            A("s_mov_b32 %s, 0x%08x" % (S_P[j], (Q >> (32 * j)) & M32))
        for j in range(8):
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.TWOQ[j], (TWOQ >> (32 * j)) & M32))
        for t in rm.Tz:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("s_mov_b64 %s, 0" % S_EXC)
        A("v_add_u32_e32 %s, -1, %%[len]" % last)                      # index of the last member: prefetches beyond it are clamped to it

        def load_xy(dst_x, dst_y):
            A("v_lshlrev_b32_e32 %s, 6, %s" % (rm.off, rm.rec))         # 64 bytes per affine point; the shift drops the sign bit (index < 2^25)
            for k, regs in enumerate((dst_x[:4], dst_x[4:], dst_y[:4], dst_y[4:])):
                A("global_load_dwordx4 %s, %s, %%[aff] offset:%d" % (quad(regs), rm.off, 16 * k))

        def neg_y(Y):
            En = Emitter()
            seq = [i_subco(rm.T2[0], rm.TWOQ[0], Y[0], "vcc")] + [i_subb(rm.T2[j], rm.TWOQ[j], Y[j], "vcc") for j in range(1, 8)]
            En.lastw[S_NEG] = -1
            seq += [i_cnd(Y[j], Y[j], rm.T2[j], S_NEG) for j in range(8)]
            En.schedule(seq)
            L.extend(En.lines)

        A("global_load_dword %s, %%[lo4], %%[vals]" % rm.rec)
        A("v_min_u32_e32 %s, 1, %s" % (rm.tmp, last))
        A("v_lshl_add_u32 %s, %s, 2, %%[lo4]" % (rm.tmp, rm.tmp))
        A("global_load_dword %s, %s, %%[vals]" % (rec2, rm.tmp))
        A("s_waitcnt vmcnt(1)")
        load_xy(rm.X1, rm.Y1)
        A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
        one = R % Q
        for j in range(8):
            A("v_mov_b32_e32 %s, 0x%08x" % (rm.Z1[j], (one >> (32 * j)) & M32))
        A("s_waitcnt vmcnt(0)")
        neg_y(rm.Y1)
        A("v_mov_b32_e32 %s, %s" % (rm.rec, rec2))
        load_xy(rm.X2, rm.Y2)
        A("s_mov_b32 %s, 1" % S_STEP)
        A("s_cmp_ge_u32 %s, %%[maxlen]" % S_STEP)
        A("s_cbranch_scc1 " + lbl("L_done"))
        align_head(A)
        A(lbl("L_step") + ":")
        A("s_add_u32 %s, %s, 1" % (S_TMP, S_STEP))
        A("v_min_u32_e32 %s, %s, %s" % (rm.tmp, S_TMP, last))
        A("v_lshl_add_u32 %s, %s, 2, %%[lo4]" % (rm.tmp, rm.tmp))
        A("global_load_dword %s, %s, %%[vals]" % (rec2, rm.tmp))
        A("v_cmp_gt_u32_e64 %s, %%[len], %s" % (S_NZ, S_STEP))          # this lane still has a member at this step
        A("v_cmp_gt_i32_e64 %s, 0, %s" % (S_NEG, rm.rec))
        for d, s_ in zip(rm.SX + rm.SY + rm.SZ, rm.X1 + rm.Y1 + rm.Z1):
            A("v_mov_b32_e32 %s, %s" % (d, s_))
        A("s_waitcnt vmcnt(1)")
        neg_y(rm.Y2)
        Ea = Emitter()
        Ea.schedule(seq_madd(rm))
        L.extend(Ea.lines)
        A("s_nop 1")
        A("s_or_b64 %s, %s, %s" % (S_M1, S_M1, S_M2))
        A("s_and_b64 %s, %s, %s" % (S_M1, S_M1, S_NZ))
        A("s_or_b64 %s, %s, %s" % (S_EXC, S_EXC, S_M1))
        for d, s_ in zip(rm.X1 + rm.Y1 + rm.Z1, rm.SX + rm.SY + rm.SZ):
            A("v_cndmask_b32_e64 %s, %s, %s, %s" % (d, s_, d, S_NZ))
        A("s_waitcnt vmcnt(0)")
        A("v_mov_b32_e32 %s, %s" % (rm.rec, rec2))
        load_xy(rm.X2, rm.Y2)
        A("s_add_u32 %s, %s, 1" % (S_STEP, S_STEP))
        A("s_cmp_lt_u32 %s, %%[maxlen]" % S_STEP)
        A("s_cbranch_scc1 " + lbl("L_step"))
        A(lbl("L_done") + ":")
        for k, regs in enumerate((rm.X1[:4], rm.X1[4:], rm.Y1[:4], rm.Y1[4:], rm.Z1[:4], rm.Z1[4:])):
            A("global_store_dwordx4 %%[dst], %s, off offset:%d" % (quad(regs), 16 * k))
        A("v_cndmask_b32_e64 %s, 0, 1, %s" % (rm.flag, S_EXC))
        A("global_store_dword %%[t4], %s, %%[exc]" % rm.flag)
        A("s_waitcnt vmcnt(0)")
        return L, rm, dict(madd=len(Ea.order), madd_nops=Ea.nops, vgpr_end=rm.end)
    return _with_globals(go)


def canon(rm, a, out, tmp):
    """[0, 2q) -> [0, q): out = a - q unless that borrows"""
    c1, _ = _carries()
    seq = [i_subco(tmp[0], a[0], rm.QV[0], c1)] + [i_subb(tmp[j], a[j], rm.QV[j], c1) for j in range(1, 8)]
    seq += [i_cnd(out[j], tmp[j], a[j], c1) for j in range(8)]
    return seq


def emit_table():
    """The window table of one scalar-mul as a hand-scheduled stream: the 16 multiples of P built with the loop's own double /
    mixed-add bodies on the curve where P is affine (P = (X, Y, Z) is the affine point (X, Y) of y^2 = x^3 + 3 Z^6), rescaled to ONE
    common Z (prefix / suffix products), plus the blinding point and its final correction mapped to that curve.
    Operands: %[tid] %[poff] (VGPR: lane index, byte offset of the lane's point), %[n] (SGPR), %[pts] %[jtab] %[tab] %[zc] (SGPR pairs)."""
    def go():
        rm = RegMap(table_kernel=True)
        L = []
        A = L.append
        lbl = lambda s_: "%s_%%=" % s_
        S_E, S_N96, S_IDX = S_STEP, "s56", "s57"
        inv = (-pow(Q, -1, 1 << 32)) & M32
        one = R % Q

        def sched(seq):
            E = Emitter()
            E.schedule(seq)
            L.extend(E.lines)
            return E

        def ld(regs, off, base, byte=0):
            A("global_load_dwordx4 %s, %s, %%[%s] offset:%d" % (quad(regs[:4]), off, base, byte))
            A("global_load_dwordx4 %s, %s, %%[%s] offset:%d" % (quad(regs[4:]), off, base, byte + 16))

        def st(regs, off, base, byte=0):
            A("global_store_dwordx4 %s, %s, %%[%s] offset:%d" % (off, quad(regs[:4]), base, byte))
            A("global_store_dwordx4 %s, %s, %%[%s] offset:%d" % (off, quad(regs[4:]), base, byte + 16))

        def entry_off(dst, sidx, stride_s, tid_v):          # dst = sidx * n * stride + tid * stride
            A("s_mul_i32 %s, %s, %s" % (S_TMP, sidx, stride_s))
            A("v_add_u32_e32 %s, %s, %s" % (dst, S_TMP, tid_v))

        def const_to(sregs, value):
            for j in range(8):
                A("s_mov_b32 %s, 0x%08x" % (sregs[j], (value >> (32 * j)) & M32))

        def mov_const(regs, value):
            for j in range(8):
                A("v_mov_b32_e32 %s, 0x%08x" % (regs[j], (value >> (32 * j)) & M32))

        A("s_nop 1")
        A("s_mov_b32 %s, 0x%08x" % (S_INV, inv))
        const_to(S_P, Q)
        mov_const(rm.TWOQ, TWOQ)
        mov_const(rm.QV, Q)
        for t in rm.Tz:
            A("v_mov_b32_e32 %s, 0" % t[1])
        A("v_lshlrev_b32_e32 %s, 5, %%[tid]" % rm.tid32)
        A("v_lshlrev_b32_e32 %s, 6, %%[tid]" % rm.tid64)
        A("v_mul_u32_u24_e32 %s, 96, %%[tid]" % rm.tid96)
        A("s_lshl_b32 %s, %%[n], 6" % S_N64)
        A("s_mul_i32 %s, %%[n], 96" % S_N96)
        # ---- T1 = P as the affine point (X, Y) of the curve scaled by its own Z; T2 = 2 T1
        ld(rm.X1, "%[poff]", "pts", 0); ld(rm.Y1, "%[poff]", "pts", 32)
        mov_const(rm.Z1, one)
        A("s_waitcnt vmcnt(0)")
        st(rm.X1, rm.tid96, "jtab", 0); st(rm.Y1, rm.tid96, "jtab", 32); st(rm.Z1, rm.tid96, "jtab", 64)
        Ed = sched(seq_double(rm))
        A("s_mov_b32 %s, 1" % S_E)
        entry_off(rm.off, S_E, S_N96, rm.tid96)
        st(rm.X1, rm.off, "jtab", 0); st(rm.Y1, rm.off, "jtab", 32); st(rm.Z1, rm.off, "jtab", 64)
        ld(rm.SY, "%[poff]", "pts", 0); ld(rm.SZ, "%[poff]", "pts", 32); ld(rm.SX, "%[poff]", "pts", 64)      # P again: (x, y) for the additions, Z for the end
        A("s_waitcnt vmcnt(0)")
        # ---- T[e+1] = T[e] + P, e = 2 .. 15 (entry index = multiple - 1)
        A("s_mov_b32 %s, 2" % S_E)
        A(lbl("T_build") + ":")
        for d, s_ in zip(rm.X2 + rm.Y2, rm.SY + rm.SZ):
            A("v_mov_b32_e32 %s, %s" % (d, s_))
        Ea = sched(seq_madd(rm))
        entry_off(rm.off, S_E, S_N96, rm.tid96)
        st(rm.X1, rm.off, "jtab", 0); st(rm.Y1, rm.off, "jtab", 32); st(rm.Z1, rm.off, "jtab", 64)
        A("s_add_u32 %s, %s, 1" % (S_E, S_E))
        A("s_cmp_lt_u32 %s, 16" % S_E)
        A("s_cbranch_scc1 " + lbl("T_build"))
        A("s_waitcnt vmcnt(0)")
        # ---- prefix products of the z's: p_e = z_0 ... z_e (z_0 = 1), p_e parked in the x slot of tab[e]; "p_-1" = 1 in the x slot of tab[17]
        mov_const(rm.X1, one)
        A("s_mov_b32 %s, 17" % S_IDX)
        entry_off(rm.off, S_IDX, S_N64, rm.tid64)
        st(rm.X1, rm.off, "tab", 0)
        st(rm.X1, rm.tid64, "tab", 0)                                  # p_0 = 1
        A("s_mov_b32 %s, 1" % S_E)
        A(lbl("T_prefix") + ":")
        entry_off(rm.off, S_E, S_N96, rm.tid96)
        ld(rm.T0, rm.off, "jtab", 64)
        A("s_waitcnt vmcnt(0)")
        sched(montmul(rm, rm.X1, rm.T0, rm.X1))
        entry_off(rm.off, S_E, S_N64, rm.tid64)
        st(rm.X1, rm.off, "tab", 0)
        A("s_add_u32 %s, %s, 1" % (S_E, S_E))
        A("s_cmp_lt_u32 %s, 16" % S_E)
        A("s_cbranch_scc1 " + lbl("T_prefix"))
        for d, s_ in zip(rm.Z1, rm.X1):                                # Zc = p_15
            A("v_mov_b32_e32 %s, %s" % (d, s_))
        A("s_waitcnt vmcnt(0)")
        # ---- backward: c_e = p_{e-1} * (z_{e+1} ... z_15) = Zc / z_e ; x' = X c^2, y' = Y c^3 ; canonical stores
        mov_const(rm.Y1, one)                                          # suffix product
        A("s_mov_b32 %s, 15" % S_E)
        A(lbl("T_back") + ":")
        A("s_sub_u32 %s, %s, 1" % (S_IDX, S_E))
        A("s_cmp_eq_u32 %s, 0" % S_E)
        A("s_cselect_b32 %s, 17, %s" % (S_IDX, S_IDX))
        entry_off(rm.off, S_IDX, S_N64, rm.tid64)
        ld(rm.T0, rm.off, "tab", 0)
        entry_off(rm.off2, S_E, S_N96, rm.tid96)
        ld(rm.X2, rm.off2, "jtab", 0); ld(rm.Y2, rm.off2, "jtab", 32); ld(rm.SZ, rm.off2, "jtab", 64)
        A("s_waitcnt vmcnt(0)")
        seq = montmul(rm, rm.T0, rm.Y1, rm.T0)                         # c
        seq += montsqr(rm, rm.T0, rm.T1)                               # c^2
        seq += montmul(rm, rm.X2, rm.T1, rm.X2)                        # x'
        seq += montmul(rm, rm.T1, rm.T0, rm.T1)                        # c^3
        seq += montmul(rm, rm.Y2, rm.T1, rm.Y2)                        # y'
        seq += montmul(rm, rm.Y1, rm.SZ, rm.Y1)                        # suffix *= z_e
        seq += canon(rm, rm.X2, rm.X2, rm.T2) + canon(rm, rm.Y2, rm.Y2, rm.T0)
        sched(seq)
        entry_off(rm.off, S_E, S_N64, rm.tid64)
        st(rm.X2, rm.off, "tab", 0); st(rm.Y2, rm.off, "tab", 32)
        A("s_waitcnt vmcnt(0)")                                        # the next iteration reads the x slot of tab[e-1] (written in the prefix pass: safe) and reuses X2 / Y2
        A("s_cmp_eq_u32 %s, 0" % S_E)
        A("s_cbranch_scc1 " + lbl("T_back_done"))
        A("s_sub_u32 %s, %s, 1" % (S_E, S_E))
        A("s_branch " + lbl("T_back"))
        A(lbl("T_back_done") + ":")
        # ---- total Z of the table on the ORIGINAL curve: Zt = Zc * Z_P; blinding point and correction on the table's curve
        seq = montmul(rm, rm.Z1, rm.SX, rm.Z1)                         # Zt
        seq += montsqr(rm, rm.Z1, rm.T0)                               # Zt^2
        seq += montmul(rm, rm.T0, rm.Z1, rm.T1)                        # Zt^3
        sched(seq)
        import hashlib
        t_ = int.from_bytes(hashlib.sha3_256(b"arkmpc g1 window-loop blinding point R0").digest(), "big") % RORD
        R0 = g1_mul(GEN, t_)
        C = g1_mul(R0, (1 << (5 * (N_STEPS // 2 - 1))) % RORD)
        for idx, (cx, cy) in ((16, R0), (17, (C[0], Q - C[1]))):
            const_to(S_BETA, mont(cx))
            sched(montmul(rm, rm.T0, S_BETA, rm.X2))
            const_to(S_BETA, mont(cy))
            sched(montmul(rm, rm.T1, S_BETA, rm.Y2) + canon(rm, rm.X2, rm.X2, rm.T2) + canon(rm, rm.Y2, rm.Y2, rm.X1))
            A("s_mov_b32 %s, %d" % (S_IDX, idx))
            entry_off(rm.off, S_IDX, S_N64, rm.tid64)
            st(rm.X2, rm.off, "tab", 0); st(rm.Y2, rm.off, "tab", 32)
            A("s_waitcnt vmcnt(0)")
        sched(canon(rm, rm.Z1, rm.Z1, rm.T2))
        st(rm.Z1, rm.tid32, "zc", 0)
        A("s_waitcnt vmcnt(0)")
        return L, rm, dict(double=len(Ed.order), madd=len(Ea.order), vgpr_end=rm.end)
    return _with_globals(go)


def emit_header(path):
    selftest(trials=24)
    lines, rm, st = emit_loop()
    out = []
    out.append("// GENERATED by tools/gen_ec_asm.py -- do not edit.  The BN254 G1 window loop as one hand-scheduled gfx950 instruction stream;")
    out.append("// see the generator for the algorithm (effective-affine table, blinded accumulator, per-lane exceptional flag) and the")
    out.append("// emulator check of the two bodies.  double: %d instructions (%d wait states), mixed add: %d (%d), VGPRs v%d..v%d." %
               (st["double"], st["double_nops"], st["madd"], st["madd_nops"], rm.first, rm.end - 1))
    out.append("#pragma once")
    import hashlib
    t = int.from_bytes(hashlib.sha3_256(b"arkmpc g1 window-loop blinding point R0").digest(), "big") % RORD
    R0 = g1_mul(GEN, t)
    C = g1_mul(R0, (1 << (5 * (N_STEPS // 2 - 1))) % RORD)            # 26 x 5 doublings are applied to the initial accumulator
    negC = (C[0], Q - C[1])
    limbs = lambda v: ", ".join("0x%08xu" % ((mont(v) >> (32 * i)) & M32) for i in range(8))
    out.append("// R0 = [SHA3-256(\"arkmpc g1 window-loop blinding point R0\") mod r] G and -(2^%d R0), affine, Montgomery form" % (5 * (N_STEPS // 2 - 1)))
    for nm, v in (("R0X", R0[0]), ("R0Y", R0[1]), ("NCX", negC[0]), ("NCY", negC[1])):
        out.append("__device__ constexpr u32 G1_ASM_%s[8] = {%s};" % (nm, limbs(v)))
    out.append("#define G1_ASM_STEPS %d" % N_STEPS)
    out.append("#define G1_ASM_TABLE %d" % N_TABLE)
    out.append("__device__ __forceinline__ void g1_smul_loop_asm(u32 tid, u32 n, u32 ptid, u32 np, const u64* tab, const u32* dig, u64* res, u32* exc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(lines))
    out.append("        :")
    out.append('        : [tid] "v"(tid), [n] "s"(n), [ptid] "v"(ptid), [np] "s"(np), [tab] "s"(tab), [dig] "s"(dig), [res] "s"(res), [exc] "s"(exc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(rm.first, rm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    tlines, trm, tst = emit_table()
    out.append("// the window table of one scalar-mul (16 multiples on the curve where P is affine, rescaled to a common Z; blinding point")
    out.append("// and correction): %d asm lines, VGPRs v%d..v%d" % (len(tlines), trm.first, trm.end - 1))
    out.append("__device__ __forceinline__ void g1_smul_table_asm(u32 tid, u32 poff, u32 n, const u64* pts, u64* jtab, u64* tab, u64* zc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(tlines))
    out.append("        :")
    out.append('        : [tid] "v"(tid), [poff] "v"(poff), [n] "s"(n), [pts] "s"(pts), [jtab] "s"(jtab), [tab] "s"(tab), [zc] "s"(zc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS + ["s56", "s57"]] + ['"v%d"' % i for i in range(trm.first, trm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    mlines, mrm, mst = emit_msm_acc()
    out.append("// bucket accumulation of the variable-base MSM on the same mixed-addition body: %d asm lines, VGPRs v%d..v%d" % (len(mlines), mrm.first, mrm.end - 1))
    out.append("__device__ __forceinline__ void g1_msm_acc_asm(u32 lo4, u32 len, u32 maxlen, const u32* vals, const u64* aff, u64* dst, u32 t4, u32* exc) {")
    out.append("    asm volatile(")
    out.append(G.c_string(mlines))
    out.append("        :")
    out.append('        : [lo4] "v"(lo4), [len] "v"(len), [maxlen] "s"(maxlen), [vals] "s"(vals), [aff] "s"(aff), [dst] "v"(dst), [t4] "v"(t4), [exc] "s"(exc)')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mrm.first, mrm.end)]
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    # multiplier instructions (v_mad_u64_u32 + v_mul_lo_u32) one scalar-mul executes in the two asm kernels
    def mults(seq_fn, *a):
        def go():
            rm = RegMap(table_kernel=True)
            return sum(1 for i in seq_fn(rm, *a) if i.op in ("mad", "mul_lo"))
        return _with_globals(go)
    m_dbl, m_madd = mults(seq_double), mults(seq_madd)
    m_mul = mults(lambda rm: montmul(rm, rm.X1, rm.Y1, rm.Z1))
    m_sqr = mults(lambda rm: montsqr(rm, rm.X1, rm.Z1))
    n_dbl = 5 * (N_STEPS // 2 - 1)
    loop_m = n_dbl * m_dbl + N_STEPS * m_madd + (N_STEPS // 2) * m_mul
    table_m = m_dbl + 14 * m_madd + 15 * m_mul + 16 * (5 * m_mul + m_sqr) + (2 * m_mul + m_sqr) + 4 * m_mul
    out.insert(4, "// multiplier instructions per scalar-mul: window loop %d (%d doublings x %d, %d mixed additions x %d, %d beta products x %d), table %d"
               % (loop_m, n_dbl, m_dbl, N_STEPS, m_madd, N_STEPS // 2, m_mul, table_m))
    out.insert(5, "#define G1_ASM_MULT_INSTRS_LOOP %d\n#define G1_ASM_MULT_INSTRS_TABLE %d" % (loop_m, table_m))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    with open(os.path.join(os.path.dirname(path), "ec_asm_stats.json"), "w") as f:
        import json
        json.dump({"mult_instrs_loop": loop_m, "mult_instrs_table": table_m, "mult_instrs_per_montmul": m_mul, "mult_instrs_per_montsqr": m_sqr,
                   "doublings": n_dbl, "mixed_additions": N_STEPS, "beta_products": N_STEPS // 2,
                   "double_body_instrs": st["double"], "madd_body_instrs": st["madd"]}, f, indent=1)
    return st, len(lines) + len(tlines) + len(mlines)


def load_beta():
    """beta (Montgomery form) from glv_consts.inc, so the loop and the compiled path use the same cube root of unity"""
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "glv_consts.inc")).read()
    m = re.search(r"GLV_BETA_MONT\[8\]\s*=\s*\{([^}]*)\}", txt)
    limbs = [int(x.strip().rstrip("u"), 0) for x in m.group(1).split(",") if x.strip()]
    return sum(l << (32 * i) for i, l in enumerate(limbs))


BETA = load_beta()
assert pow(unmont(BETA), 3, Q) == 1 and unmont(BETA) != 1

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark-mpc_amd", "csrc", "ec_asm_kernels.inc"))
    a = ap.parse_args()
    if a.selftest:
        Ed, Ea = selftest(trials=200)
        print("double ok: %d instructions, %d wait states; madd ok: %d instructions, %d wait states" % (len(Ed.order), Ed.nops, len(Ea.order), Ea.nops))
        sys.exit(0)
    st, nlines = emit_header(a.o)
    print("ec loop: %d asm lines; %s" % (nlines, st))
