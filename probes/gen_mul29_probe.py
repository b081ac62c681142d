#!/usr/bin/env python3
"""Gate experiment for the BN254 scalar-mul loop: a HAND-SCHEDULED Montgomery multiplication on unsaturated 29-bit limbs (9 limbs, R = 2^261),
emitted as an inline-asm block for probes/mulrate29.hip.  Product scanning: every column accumulates in a 64-bit VGPR pair through the addend
of v_mad_u64_u32 (no carry folds), the reduction term m_k q_0 clears the low 29 bits, one v_lshrrev_b64 carries the column into the next.
Two accumulators per column (even / odd terms) break the dependent chain.  The stream is executed by the single-lane emulator against Python
integers before it is written (selftest below).

usage: python probes/gen_mul29_probe.py  ->  probes/mul29_asm.inc"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import gen_asm_kernels as G
from gen_asm_kernels import Ins, Emitter, Emu, M32, regs_of, pr, src

Q = dict(G.FIELDS)["BN254_FQ"]
M29 = (1 << 29) - 1
RP = 1 << 261
S_JUNK, S_INV, S_MASK = "s[16:17]", "s18", "s19"
S_Q = ["s%d" % (20 + i) for i in range(9)]
CLOBBER_SGPRS = ["s%d" % i for i in range(16, 29)]


def i_mad(d, a, b, c):
    return Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(d), S_JUNK, a, b, "0" if c == 0 else pr(c)), "mad", (d, a, b, c), rd=regs_of(a, b, c if c != 0 else None), wr=list(d))
def i_mul_lo(d, a, b): return Ins("v_mul_lo_u32 %s, %s, %s" % (d, a, b), "mul_lo", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_and(d, a, b): return Ins("v_and_b32_e32 %s, %s, %s" % (d, src(a), b), "and", (d, a, b), rd=regs_of(a, b), wr=[d])
def i_shr64(d, s_, sh): return Ins("v_lshrrev_b64 %s, %d, %s" % (pr(d), sh, pr(s_)), "shr64", (d, s_, sh), rd=list(s_), wr=list(d))
def i_add64(d, a, b): return Ins("v_lshl_add_u64 %s, %s, 0, %s" % (pr(d), pr(a), pr(b)), "add64", (d, a, b), rd=list(a) + list(b), wr=list(d))
def i_mov(d, s_): return Ins("v_mov_b32_e32 %s, %s" % (d, src(s_)), "mov", (d, s_), rd=regs_of(s_), wr=[d])
def i_alignbit(d, hi, lo, sh): return Ins("v_alignbit_b32 %s, %s, %s, %d" % (d, hi, lo, sh), "alignbit", (d, hi, lo, sh), rd=regs_of(hi, lo), wr=[d])
def i_lshr(d, a, sh): return Ins("v_lshrrev_b32_e32 %s, %d, %s" % (d, sh, a), "lshr", (d, a, sh), rd=regs_of(a), wr=[d])
SPLIT_SHIFT = [False]        # the column carry as v_alignbit_b32 + v_lshrrev_b32 instead of one v_lshrrev_b64
def carry_shift(acc):
    if SPLIT_SHIFT[0]:
        return [i_alignbit(acc[0], acc[1], acc[0], 29), i_lshr(acc[1], acc[1], 29)]
    return [i_shr64(acc, acc, 29)]


class Emu29(Emu):
    def run(self, order):
        for ins in order:
            if ins.op == "shr64":
                d, s_, sh = ins.args
                v = (self.rd(s_[0]) | (self.rd(s_[1]) << 32)) >> sh
                self.v[d[0]], self.v[d[1]] = v & M32, v >> 32
            elif ins.op == "alignbit":
                d, hi, lo, sh = ins.args
                self.v[d] = (((self.rd(hi) << 32) | self.rd(lo)) >> sh) & M32
            elif ins.op == "lshr":
                self.v[ins.args[0]] = self.rd(ins.args[1]) >> ins.args[2]
            elif ins.op == "add64":
                d, a, b = ins.args
                v = ((self.rd(a[0]) | (self.rd(a[1]) << 32)) + (self.rd(b[0]) | (self.rd(b[1]) << 32))) & ((1 << 64) - 1)
                self.v[d[0]], self.v[d[1]] = v & M32, v >> 32
            else:
                Emu.run(self, [ins])


def mul29_seq(a, b, o, acc0, acc1, m, two_acc=True):
    """o = a * b / 2^261 mod q, limbs of o below 2^29 (top limb: the rest).  a, b: 9 registers each (limbs <= 2^30).  acc0 / acc1: 64-bit pairs."""
    seq = []
    carry = None                      # pair holding the carry into this column, or None
    for k in range(17):
        terms = [(a[i], b[k - i]) for i in range(9) if 0 <= k - i < 9]
        red = [(m[i], S_Q[k - i]) for i in range(9) if 0 <= k - i < 9 and i < min(k, 9)] if k < 9 else [(m[i], S_Q[k - i]) for i in range(9) if 0 <= k - i < 9]
        allt = terms + red
        # accumulator 0 starts from the carry, accumulator 1 from zero
        cur = [carry, 0]
        pairs = [acc0, acc1]
        for n_, (x, y) in enumerate(allt):
            w = n_ % 2 if (two_acc and len(allt) > 2) else 0
            seq.append(i_mad(pairs[w], x, y, cur[w] if cur[w] is not None else 0))
            cur[w] = pairs[w]
        if cur[1] != 0:
            seq.append(i_add64(acc0, acc0, acc1))
        if k < 9:
            seq.append(i_mul_lo(m[k], acc0[0], S_INV))
            seq.append(i_and(m[k], S_MASK, m[k]))
            seq.append(i_mad(acc0, m[k], S_Q[0], acc0))
            seq += carry_shift(acc0)
        else:
            seq.append(i_and(o[k - 9], S_MASK, acc0[0]))
            seq += carry_shift(acc0)
            if k == 16:
                seq.append(i_mov(o[8], acc0[0]))
        carry = acc0
    return seq


def build(two_acc=True):
    a = ["%%[a%d]" % i for i in range(9)]
    b = ["%%[b%d]" % i for i in range(9)]
    o = ["%%[o%d]" % i for i in range(9)]
    # fixed temporaries in caller-saved blocks
    acc0, acc1 = ("v32", "v33"), ("v34", "v35")
    m = ["v%d" % r for r in (36, 37, 38, 39, 48, 49, 50, 51, 52)]
    E = Emitter()
    E.raw("s_nop 1")
    inv = (-pow(Q, -1, 1 << 29)) % (1 << 29)
    E.raw("s_mov_b32 %s, 0x%08x" % (S_INV, inv), "smov", (S_INV, inv))
    E.raw("s_mov_b32 %s, 0x%08x" % (S_MASK, M29), "smov", (S_MASK, M29))
    for j in range(9):
        E.raw("s_mov_b32 %s, 0x%08x" % (S_Q[j], (Q >> (29 * j)) & M29 if j < 8 else Q >> 232), "smov", (S_Q[j], (Q >> (29 * j)) & M29 if j < 8 else Q >> 232))
    E.schedule(mul29_seq(a, b, o, acc0, acc1, m, two_acc))
    used = [32, 33, 34, 35, 36, 37, 38, 39, 48, 49, 50, 51, 52]
    return E, dict(a=a, b=b, o=o, used=used)


def selftest(trials=300, two_acc=True):
    rng = random.Random(9)
    E, mp = build(two_acc)
    Rinv = pow(RP, -1, Q)
    for t in range(trials):
        if t < 20:
            x = [rng.choice([0, 1, M29, (1 << 30) - 1]) for _ in range(9)]
            y = [rng.choice([0, 1, M29, (1 << 30) - 1]) for _ in range(9)]
        else:
            x = [rng.randrange(1 << 30) for _ in range(9)]
            y = [rng.randrange(1 << 30) for _ in range(9)]
        if t == 0:
            x = [(1 << 30) - 1] * 9; y = [(1 << 30) - 1] * 9
        em = Emu29()
        for i in range(9):
            em.s[mp["a"][i]] = x[i]; em.s[mp["b"][i]] = y[i]
        em.run(E.order)
        got = [em.v[mp["o"][i]] for i in range(9)]
        X = sum(v << (29 * i) for i, v in enumerate(x)); Y = sum(v << (29 * i) for i, v in enumerate(y))
        G_ = sum(v << (29 * i) for i, v in enumerate(got))
        assert G_ % Q == X * Y * Rinv % Q, t
        assert all(v <= M29 for v in got[:8]), t
        assert G_ < (X * Y) // RP + Q + 1
    return E, mp


def main():
    out = []
    out.append("// GENERATED by probes/gen_mul29_probe.py -- do not edit.  Hand-scheduled 29-bit-limb Montgomery multiplication (BN254 Fq, R = 2^261) for probes/mulrate29.hip.")
    for name, two in (("mul29_asm", True), ("mul29_asm_1acc", False), ("mul29_asm_1acc_split", False)):
        SPLIT_SHIFT[0] = name.endswith("_split")
        E, mp = selftest(two_acc=two)
        nmad = sum(1 for i in E.order if i.op == "mad")
        nother = sum(1 for i in E.order if i.op in ("mul_lo", "and", "shr64", "add64", "mov", "alignbit", "lshr"))
        clob = ['"vcc"'] + ['"%s"' % s_ for s_ in CLOBBER_SGPRS] + ['"v%d"' % i for i in mp["used"]]
        out.append("// %s: %d v_mad_u64_u32 + %d other VALU" % (name, nmad, nother))
        out.append("__device__ __forceinline__ F29 %s(const F29& a, const F29& b) {" % name)
        out.append("    F29 o;")
        out.append("    asm(")
        out.append(G.c_string(E.lines))
        out.append("        : " + ", ".join('[o%d] "=&v"(o.v[%d])' % (i, i) for i in range(9)))
        out.append("        : " + ", ".join('[a%d] "v"(a.v[%d])' % (i, i) for i in range(9)) + ",")
        out.append("          " + ", ".join('[b%d] "v"(b.v[%d])' % (i, i) for i in range(9)))
        out.append("        : " + ", ".join(clob) + ");")
        out.append("    return o;")
        out.append("}")
        print("%s: %d mad, %d other, selftest ok" % (name, nmad, nother))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "probes", "mul29_asm.inc")
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
