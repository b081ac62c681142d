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
]
M32 = 0xFFFFFFFF
R = 1 << 256
JUNK = "s[60:61]"     # carry-out sink of v_mad_u64_u32 (never read)
S_INV = "s62"         # -p^{-1} mod 2^32
CLOBBER_SGPRS = ["s60", "s61", "s62"]


class Ins:
    __slots__ = ("text", "op", "args", "srd", "swr")

    def __init__(self, text, op, args=(), srd=(), swr=()):
        self.text, self.op, self.args, self.srd, self.swr = text, op, args, tuple(srd), tuple(swr)


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


# ---- instruction constructors (VCC is the only carry register used) -----------------------------
def i_mov(d, s): return Ins("v_mov_b32_e32 %s, %s" % (d, src(s)), "mov", (d, s))
def i_mad(d, a, b, c): return Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(d), JUNK, a, b, "0" if c == 0 else pr(c)), "mad", (d, a, b, c))
def i_mul_lo(d, a, b): return Ins("v_mul_lo_u32 %s, %s, %s" % (d, a, b), "mul_lo", (d, a, b))
def i_addco(d, a, b): return Ins("v_add_co_u32_e32 %s, vcc, %s, %s" % (d, src(a), b), "addco", (d, a, b), swr=("vcc",))
def i_addc(d, a, b): return Ins("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (d, src(a), b), "addc", (d, a, b), srd=("vcc",), swr=("vcc",))
def i_subco(d, a, b): return Ins("v_sub_co_u32_e32 %s, vcc, %s, %s" % (d, src(a), b), "subco", (d, a, b), swr=("vcc",))
def i_subb(d, a, b): return Ins("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (d, src(a), b), "subb", (d, a, b), srd=("vcc",), swr=("vcc",))
def i_cnd(d, f, t): return Ins("v_cndmask_b32_e32 %s, %s, %s, vcc" % (d, f, t), "cnd", (d, f, t), srd=("vcc",))   # vcc ? t : f
def i_and(d, a, b): return Ins("v_and_b32_e32 %s, %s, %s" % (d, src(a), b), "and", (d, a, b))


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


def montmul_sum_rows(p, prods, T, Tz, q, m, P):
    """Rows of the CIOS evaluation of REDC(sum_k a_k*b_k).  T[j] is the low half of the pair Tz[j] whose high
    half holds 0 for the whole kernel.  Each row = (pre, mads[8], chain); chain element i (0-based) is the
    instruction after which the NEXT row's mad_i may be issued (it has produced T_i and released q_i)."""
    nprod = len(prods)
    before, _ = t_bounds(p, nprod)
    assert before < (1 << 288), "accumulator needs a 10th limb for this field / product count"
    rows = []
    first = True
    for r in range(8):
        for (a, b) in prods:
            mads = [i_mad(q[j], a[j], b[r], 0 if first else Tz[j]) for j in range(8)]
            chain = [i_mov(T[0], q[0][0]), i_addco(T[1], q[1][0], q[0][1])]
            chain += [i_addc(T[j], q[j][0], q[j - 1][1]) for j in range(2, 8)]
            chain += [i_addc(T[8], 0 if first else T[8], q[7][1])]
            # release points: next.mad_j needs T_j (chain[j]) and q_j free (read by chain[j], chain[j+1])
            rel = [min(j + 1, 8) for j in range(8)]
            rows.append(("mul", [], mads, chain, rel, 0))
            first = False
        pre = [i_mul_lo(m, T[0], S_INV)]
        mads = [i_mad(q[j], m, P[j], Tz[j]) for j in range(8)]
        chain = [i_addco(T[0], q[1][0], q[0][1])]
        chain += [i_addc(T[j - 1], q[j][0], q[j - 1][1]) for j in range(2, 8)]
        chain += [i_addc(T[7], T[8], q[7][1]), i_addc(T[8], 0, Tz[8][1])]   # Tz[8][1] holds 0 (VOP2 src1 must be a VGPR)
        # chain[i] writes T_i (i = 0..7); q_j is read by chain[j-1] (lo) and chain[j] (hi)
        rel = [min(j, 7) for j in range(8)]
        rows.append(("red", pre, mads, chain, rel, 0))
    return rows


def pipeline_rows(rows):
    """Flatten rows, issuing row k+1's multiplies inside row k's carry chain (next.mad_j right after the
    chain element that releases it).  `pre` of the next row (the m = T0*inv multiply) needs the new T_0."""
    seq = []
    seq += rows[0][1] + rows[0][2]
    for k, row in enumerate(rows):
        chain, rel = row[3], row[4]
        nxt = rows[k + 1] if k + 1 < len(rows) else None
        if nxt is None:
            seq += chain
            break
        npre, nmads, nrel_by_mad = nxt[1], nxt[2], rel
        pending = list(range(8))
        pre_done = False
        for ci, c in enumerate(chain):
            seq.append(c)
            if not pre_done and ci >= 0:
                seq += npre          # T_0 is produced by chain[0] in both row kinds
                pre_done = True
            while pending and nrel_by_mad[pending[0]] <= ci:
                seq.append(nmads[pending.pop(0)])
                break                # at most one multiply per chain link keeps the links evenly spaced
        for j in pending:
            seq.append(nmads[j])
    return seq


def cond_sub_final(p, nprod, T, P, tmp, out):
    """T (9 limbs) < p*(1 + nprod*p/R) + 1 -> canonical `out` by K conditional subtractions."""
    bound = (nprod * (p - 1) * (p - 1) + (R - 1) * p) // R + 1
    K = (bound + p - 1) // p - 1
    nine = bound >= R
    seq = []
    cur = T
    for k in range(K):
        last = k == K - 1
        dst = out if last else T
        seq.append(i_subco(tmp[0], cur[0], P[0]))
        seq += [i_subb(tmp[j], cur[j], P[j]) for j in range(1, 8)]
        if nine:
            seq.append(i_subb(tmp[8], cur[8], "v_zero"))
        seq += [i_cnd(dst[j], tmp[j], cur[j]) for j in range(8)]          # borrow -> value < p -> keep cur
        if nine and not last:
            seq.append(i_cnd(T[8], tmp[8], cur[8]))
        cur = dst
    if K == 0:
        seq += [i_mov(out[j], T[j]) for j in range(8)]
    return seq


def fe_add_seq(P, a, b, out, tmp):
    """out = a + b mod p for canonical a, b (p < 2^255: no 257th bit). `out` may alias a or b; tmp may not."""
    seq = [i_addco(tmp[0], a[0], b[0])] + [i_addc(tmp[j], a[j], b[j]) for j in range(1, 8)]
    seq += [i_subco(out[0], tmp[0], P[0])] + [i_subb(out[j], tmp[j], P[j]) for j in range(1, 8)]
    seq += [i_cnd(out[j], out[j], tmp[j]) for j in range(8)]
    return seq


# ------------------------------------------------------------------------------------------------
# single-lane emulator
# ------------------------------------------------------------------------------------------------
class Emu:
    def __init__(self):
        self.v, self.s, self.vcc = {}, {}, 0

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
                r = self.rd(a[1]) + self.rd(a[2]) + (self.vcc if op == "addc" else 0)
                self.v[a[0]], self.vcc = r & M32, r >> 32
            elif op in ("subco", "subb"):
                r = self.rd(a[1]) - self.rd(a[2]) - (self.vcc if op == "subb" else 0)
                self.v[a[0]], self.vcc = r & M32, 1 if r < 0 else 0
            elif op == "cnd":
                self.v[a[0]] = self.rd(a[2]) if self.vcc else self.rd(a[1])
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
KEY_S = ["%%[k%d]" % i for i in range(8)]


def build_beaver_finish(p, first_vgpr=8, key_names=None):
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
    E.raw("s_nop 1")                                                  # H1 guard for SGPR inputs written by VALU
    E.raw("s_mov_b32 %s, 0x%08x" % (S_INV, inv), "smov", (S_INV, inv))
    # ---- all first-wave loads up front (16 x dwordx4), in the order they are needed
    loads = [(dm, "my_d"), (dp, "peer_d"), (em, "my_e"), (ep, "peer_e")]
    cols = [(bs, "b_s"), (as_, "a_s"), (bm, "b_m"), (am, "a_m")]
    for regs, nm in loads:
        for h in (0, 1):
            E.emit(Ins("global_load_dwordx4 %s, %%[off_de], %%[%s]%s" % (quad(regs[4 * h:4 * h + 4]), nm, " offset:16" if h else ""), "load", (regs[4 * h:4 * h + 4], nm, h)))
    for regs, nm in cols:
        for h in (0, 1):
            E.emit(Ins("global_load_dwordx4 %s, %%[off_col], %%[%s]%s" % (quad(regs[4 * h:4 * h + 4]), nm, " offset:16" if h else ""), "load", (regs[4 * h:4 * h + 4], nm, h)))
    # constants while the loads fly
    for j in range(8):
        E.emit(i_mov(P[j], (p >> (32 * j)) & M32))
    for t in Tz:
        E.emit(i_mov(t[1], 0))
    # ---- K2: d = my_d + peer_d, e = my_e + peer_e
    E.raw("s_waitcnt vmcnt(8)", "wait")
    E.emit_all(fe_add_seq(P, dm, dp, dm, qflat))
    E.emit_all(fe_add_seq(P, em, ep, em, qflat))
    d, e, de = dm, em, dp
    # ---- de = d*e
    E.emit_all(pipeline_rows(montmul_sum_rows(p, [(d, e)], T, Tz, q, m, P)))
    E.emit_all(cond_sub_final(p, 1, T, P, qflat, de))
    # ---- share' = d*b.s + e*a.s  (one reduction)
    E.raw("s_waitcnt vmcnt(4)", "wait")
    E.emit_all(pipeline_rows(montmul_sum_rows(p, [(d, bs), (e, as_)], T, Tz, q, m, P)))
    rs = ep
    E.emit_all(cond_sub_final(p, 2, T, P, qflat, rs))
    # ---- c is loaded late, over the now-dead b.s / a.s registers; latency hides under the MAC rows
    cs, cm = bs, as_
    for regs, nm in ((cs, "c_s"), (cm, "c_m")):
        for h in (0, 1):
            E.emit(Ins("global_load_dwordx4 %s, %%[off_col], %%[%s]%s" % (quad(regs[4 * h:4 * h + 4]), nm, " offset:16" if h else ""), "load", (regs[4 * h:4 * h + 4], nm, h)))
    # ---- mac' = d*b.m + e*a.m + key*de  (one reduction)
    E.raw("s_waitcnt vmcnt(4)", "wait")
    E.emit_all(pipeline_rows(montmul_sum_rows(p, [(d, bm), (e, am), (key, de)], T, Tz, q, m, P)))
    rm = bm
    E.emit_all(cond_sub_final(p, 3, T, P, qflat, rm))
    # ---- share = share' + c.s + (PARTY0 ? de : 0) ; mac = mac' + c.m
    E.raw("s_waitcnt vmcnt(0)", "wait")
    dem = am
    for j in range(8):
        E.emit(i_and(dem[j], "%[mask]", de[j]))
    E.emit_all(fe_add_seq(P, rs, cs, rs, qflat))
    E.emit_all(fe_add_seq(P, rs, dem, rs, qflat))
    E.emit_all(fe_add_seq(P, rm, cm, rm, qflat))
    for regs, nm in ((rs, "out_s"), (rm, "out_m")):
        for h in (0, 1):
            E.emit(Ins("global_store_dwordx4 %%[off_out], %s, %%[%s]%s" % (quad(regs[4 * h:4 * h + 4]), nm, " offset:16" if h else ""), "store", (regs[4 * h:4 * h + 4], nm, h)))
    regmap = dict(P=P, dm=dm, dp=dp, em=em, ep=ep, bs=bs, as_=as_, bm=bm, am=am, rs=rs, rm=rm, first=first_vgpr, nv=nv)
    return E, regmap


def selftest_finish(p, trials=40, seed=1):
    rng = random.Random(seed)
    E, mp = build_beaver_finish(p, key_names=["s%d" % (70 + i) for i in range(8)])
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
    out.append("#pragma once")
    stats = []
    for fid, (name, p) in enumerate(FIELDS):
        try:
            selftest_finish(p, trials=16, seed=fid)           # emulator check (uses placeholder SGPR names for the key)
            E, mp = build_beaver_finish(p)                      # same stream with the asm operand names %[k0]..%[k7]
        except AssertionError as ex:
            if "10th limb" in str(ex):
                out.append("// %s: lazy 3-product reduction needs a 10th accumulator limb -> no asm body, C++ kernel is used" % name)
                out.append("template <> struct HasAsmFinish<%d> { static constexpr bool value = false; };" % fid)
                continue
            raise
        nvalu = sum(1 for i in E.order if i.op in ("mov", "mad", "mul_lo", "addco", "addc", "subco", "subb", "cnd", "and"))
        nmad = sum(1 for i in E.order if i.op == "mad")
        stats.append((name, nvalu, nmad, E.nops, mp["nv"]))
        clob = ['"memory"', '"vcc"'] + ['"%s"' % s for s in CLOBBER_SGPRS] + ['"v%d"' % i for i in range(mp["first"], mp["nv"])]
        out.append("// %s: %d VALU (%d v_mad_u64_u32), %d H1 wait states, VGPRs v%d..v%d" % (name, nvalu, nmad, E.nops, mp["first"], mp["nv"] - 1))
        out.append("template <> struct HasAsmFinish<%d> { static constexpr bool value = true; };" % fid)
        out.append("template <> __device__ __forceinline__ void beaver_finish_asm<%d>(u32 off_de, u32 off_col, u32 off_out, const u64* my_d, const u64* my_e," % fid)
        out.append("        const u64* peer_d, const u64* peer_e, const u64* a_s, const u64* a_m, const u64* b_s, const u64* b_m, const u64* c_s, const u64* c_m,")
        out.append("        u64* out_s, u64* out_m, const Fe& key, u32 mask) {")
        out.append("    asm volatile(")
        text = [ln.replace("v_zero", "v%d" % (mp["nv"])) for ln in E.lines]
        out.append(c_string(text))
        out.append("        :")
        out.append('        : [off_de] "v"(off_de), [off_col] "v"(off_col), [off_out] "v"(off_out), [my_d] "s"(my_d), [my_e] "s"(my_e),')
        out.append('          [peer_d] "s"(peer_d), [peer_e] "s"(peer_e), [a_s] "s"(a_s), [a_m] "s"(a_m), [b_s] "s"(b_s), [b_m] "s"(b_m),')
        out.append('          [c_s] "s"(c_s), [c_m] "s"(c_m), [out_s] "s"(out_s), [out_m] "s"(out_m), [mask] "s"(mask),')
        out.append("          " + ", ".join('[k%d] "s"(key.v[%d])' % (i, i) for i in range(8)))
        out.append("        : " + ", ".join(clob) + ");")
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
                E, mp = selftest_finish(p, trials=60)
                print("%-14s ok: %d instrs, %d wait states, %d VGPRs" % (name, len(E.order), E.nops, mp["nv"]))
            except AssertionError as ex:
                print("%-14s %s" % (name, ex))
        sys.exit(0)
    for s in emit_header(a.o):
        print("%-14s VALU %d  mad %d  H1 wait states %d  VGPR end %d" % s)
