"""CPU checks of the hand-scheduled kernel generator (tools/gen_asm_kernels.py): every emitted instruction stream is
executed by the single-lane emulator against Python big-int arithmetic, the hazard rule H1 is re-verified on the final
text, and the committed asm_kernels.inc must be exactly what the generator produces."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_asm_kernels as g  # noqa: E402


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_emulated_stream_matches_bigint(fid):
    name, p = g.FIELDS[fid]
    E, mp = g.selftest_finish(p, trials=24, seed=100 + fid)      # asserts inside on any mismatch (edge + random operands)
    assert mp["nv"] <= 128                                        # 4 waves/SIMD budget


@pytest.mark.parametrize("fid", [0, 1, 2, 3, 4])
def test_single_montmul_block_matches_bigint(fid):
    name, p = g.FIELDS[fid]
    E, mp = g.selftest_montmul(p, trials=120, seed=fid)
    assert E.nops <= 8 and len(mp["used"]) == 35
    assert not any(v >= 40 and (v - 40) % 16 < 8 for v in mp["used"])      # no callee-saved VGPRs (v40-47, v56-63, ...)


@pytest.mark.parametrize("fid", [0, 1])
def test_h1_hazard_distance_in_final_text(fid):
    """Re-derive H1 from the emitted text alone: between a VALU that writes VCC / the second carry pair and a VALU that
    reads it there must be >= 2 wait states (instructions or s_nop cycles)."""
    name, p = g.FIELDS[fid]
    E, _ = g.build_beaver_finish(p, nt=True)
    slot, lastw = 0, {}
    for ln in E.lines:
        m = re.match(r"s_nop (\d+)", ln)
        if m:
            slot += int(m.group(1)) + 1
            continue
        for cy in ("vcc", g.CY2):
            ops = ln.split(None, 1)[1] if " " in ln else ""
            reads = (ln.startswith(("v_addc", "v_subb")) and ops.rstrip().endswith(cy)) or (ln.startswith("v_cndmask") and ops.rstrip().endswith(cy))
            if reads and cy in lastw:
                assert slot - lastw[cy] >= 3, (ln, slot, lastw[cy])
        for cy in ("vcc", g.CY2):
            if re.match(r"v_(add|sub|addc|subb)_co_u32_e(32|64) v\d+, %s," % re.escape(cy), ln):
                lastw[cy] = slot
        slot += 1


def test_committed_header_is_current(tmp_path):
    out = tmp_path / "asm_kernels.inc"
    g.emit_header(str(out))
    committed = open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "asm_kernels.inc")).read()
    assert out.read_text() == committed, "regenerate with: python tools/gen_asm_kernels.py"


def test_montmul_block_lazy_range_bn254_fq():
    """Inputs anywhere in [0, 2q) give an output in [0, 2q): the range the BN254 point formulas keep coordinates in."""
    name, p = g.FIELDS[3]
    assert name == "BN254_FQ" and g.selftest_montmul_lazy(p)
    with pytest.raises(AssertionError):
        g.selftest_montmul_lazy(g.FIELDS[4][1], trials=1)          # 2^255 - 19: 4q exceeds 2^256, no lazy range there


# ---- round 2: the BN254 G1 scalar-mul streams (tools/gen_ec_asm.py -> csrc/ec_asm_kernels.inc) ----------------------------
import gen_ec_asm as ec  # noqa: E402


def test_ec_double_and_mixed_add_bodies_match_the_affine_group_law():
    """The two bodies of the window loop, scheduled exactly as they are emitted, run on the single-lane emulator for lazy-range
    Jacobian inputs and compared with the affine group law in Python integers; the exceptional inputs of the mixed addition
    (same point, opposite point) must raise the H = 0 masks instead."""
    Ed, Ea = ec.selftest(trials=48, seed=20260929)
    assert len(Ed.order) < 2100 and len(Ea.order) < 3400          # the instruction budget of the measured kernels


def test_ec_squaring_rows_and_lazy_ops_against_integers():
    import random
    rng = random.Random(5)
    Q, R, M32 = ec.Q, ec.R, ec.M32

    def run(seq_fn, setup, out_regs):
        def go():
            rm = ec.RegMap(table_kernel=True)
            E = ec.Emitter(); E.schedule(seq_fn(rm))
            em = ec._emu_for(rm)
            em.setv(rm.QV, Q)
            setup(em, rm)
            em.run(E.order)
            return em.getv(out_regs(rm))
        return ec._with_globals(go)

    Rinv = pow(R, -1, Q)
    for t in range(60):
        a = rng.choice([0, 1, Q - 1, Q, Q + 1, 2 * Q - 1]) if t < 12 else rng.randrange(2 * Q)
        b = rng.randrange(2 * Q)
        sq = run(lambda rm: ec.montsqr(rm, rm.X1, rm.Z1), lambda em, rm: em.setv(rm.X1, a), lambda rm: rm.Z1)
        assert sq < 2 * Q and sq % Q == a * a * Rinv % Q
        h = run(lambda rm: ec.half_lz(rm, rm.X1, rm.Y1, rm.T0), lambda em, rm: em.setv(rm.X1, a), lambda rm: rm.Y1)
        assert h < 2 * Q and (2 * h - a) % Q == 0
        s_ = run(lambda rm: ec.sub_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.T0), lambda em, rm: (em.setv(rm.X1, a), em.setv(rm.Y1, b)), lambda rm: rm.Z1)
        assert s_ < 2 * Q and (s_ - (a - b)) % Q == 0
        ad = run(lambda rm: ec.add_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.T0), lambda em, rm: (em.setv(rm.X1, a), em.setv(rm.Y1, b)), lambda rm: rm.Z1)
        assert ad < 2 * Q and (ad - (a + b)) % Q == 0
        c = run(lambda rm: ec.canon(rm, rm.X1, rm.Z1, rm.T0), lambda em, rm: em.setv(rm.X1, a), lambda rm: rm.Z1)
        assert c == a % Q


def test_committed_ec_header_is_current(tmp_path):
    out = tmp_path / "ec_asm_kernels.inc"
    ec.emit_header(str(out))
    committed = open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec_asm_kernels.inc")).read()
    assert out.read_text() == committed, "regenerate with: python tools/gen_ec_asm.py"
    import json
    st = json.load(open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec_asm_stats.json")))
    assert st["mult_instrs_per_montmul"] == 136 and st["mult_instrs_per_montsqr"] == 108 and st["doublings"] == 130 and st["mixed_additions"] == 55


# ---- round 3: the same two streams on nine 29-bit limbs (tools/gen_ec29_asm.py -> csrc/ec29_asm_kernels.inc) -----------------------
import gen_ec29_asm as ec29  # noqa: E402


def test_ec29_bodies_match_the_affine_group_law_and_keep_their_bounds():
    """Double and mixed addition on unsaturated limbs, scheduled as emitted, against the affine group law; outputs must satisfy the
    accumulator invariant the generator carried (limbs, value); the exceptional inputs must leave Z = 0 (mod q) and Z must stay 0
    through a following double / add (the loop's single end-of-run test relies on that)."""
    Ed, Ea = ec29.selftest(trials=48, seed=20260929)
    assert len(Ed.order) < 1600 and len(Ea.order) < 2400


def test_ec29_every_multiplication_at_its_operand_bounds():
    assert ec29.selftest_extremes() >= 15


def test_ec29_limb_operations_against_integers():
    import random
    rng = random.Random(3)
    Q, M29 = ec29.Q, ec29.M29
    rm = ec29.RegMap()
    for t in range(40):
        la = [rng.randrange(M29 + 9) for _ in range(8)] + [rng.randrange(1 << 26)]
        lb = [rng.randrange(4 * M29) for _ in range(8)] + [rng.randrange(1 << 25)]
        if t == 0:
            la = [M29 + 8] * 8 + [(1 << 26) - 1]; lb = [4 * M29] * 8 + [(1 << 25) - 1]
        a = ec29.FV(rm.X1, M29 + 8, (1 << 26) - 1, ec29.val29([M29 + 8] * 8 + [(1 << 26) - 1]))
        b = ec29.FV(rm.X2, 4 * M29, (1 << 25) - 1, ec29.val29([4 * M29] * 8 + [(1 << 25) - 1]))
        B = ec29.Bld(rm)
        d = B.sub(a, b, rm.T0)
        dn = B.norm(d, rm.T1)
        ng = B.neg(b, rm.T2)
        sa = B.shl_add(a, 1, a, rm.SY)
        E = ec29.Emitter(); E.schedule(B.seq)
        em = ec29.emu_for(); em.set9(rm.X1, la); em.set9(rm.X2, lb)
        em.run(E.order)
        va, vb = ec29.val29(la), ec29.val29(lb)
        got_d = [em.v[r] for r in rm.T0]
        assert (ec29.val29(got_d) - (va - vb)) % Q == 0 and ec29.val29(got_d) <= d.vmax and max(got_d[:8]) <= d.lmax and got_d[8] <= d.tmax
        got_n = [em.v[r] for r in rm.T1]
        assert ec29.val29(got_n) == ec29.val29(got_d) and max(got_n[:8]) <= dn.lmax <= M29 + 8
        assert (em.get9(rm.T2) + vb) % Q == 0 and em.get9(rm.T2) <= ng.vmax
        assert em.get9(rm.SY) == 3 * va


def test_committed_ec29_header_is_current(tmp_path):
    out = tmp_path / "ec29_asm_kernels.inc"
    ec29.emit_header(str(out))
    committed = open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec29_asm_kernels.inc")).read()
    assert out.read_text() == committed, "regenerate with: python tools/gen_ec29_asm.py"
    import json
    st = json.load(open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ec29_asm_stats.json")))
    assert st["mult_instrs_per_mul"] == 171 and st["mult_instrs_per_sqr"] == 135 and st["doublings"] == 130 and st["mixed_additions"] == 55


# ---- round 2: the Curve25519 window loop (tools/gen_ed_asm.py -> csrc/ed_asm_kernels.inc) ------------------------------------
import gen_ed_asm as ed  # noqa: E402


def test_ed_double_and_add_bodies_match_the_affine_edwards_law():
    """dbl-2008-hwcd (rearranged without negations) and add-2008-hwcd-3 with a cached operand, with and without the T output, on the
    emulator against the affine twisted-Edwards law in Python integers -- incl. the identity entry, P + P and P + (-P): the law is complete."""
    r = ed.selftest(trials=36, seed=20260930)
    assert len(r[False][0].order) < 1300 and len(r[False][1].order) < 1400          # plain products: about half of the Montgomery bodies


def test_ed_value_range_at_its_edges():
    """Every value the loop holds is below 2^255 + 19 * 77 (a folded 512-bit product: 2^256 = 38, 2^255 = 19 mod q); sums and differences
    come out below 2^255 + 57.  Plain products, squarings (doubled-operand rows, operand folded in place), additions and subtractions are
    exercised at the edges of that range -- where the ninth limb, the 257th bit and the top bit fire -- not just on random operands."""
    import random
    rng = random.Random(9)
    Q, LIM, B = ed.Q, ed.LIM, ed.B255
    assert LIM == B + 19 * 77 and 2 * Q > LIM
    assert ed.selftest_field(trials=120, seed=77)

    def run(seq_fn, vals, out):
        def go():
            rm = ed.RegMap()
            E = ed.Emitter(); E.schedule(seq_fn(rm))
            em = ed._emu(rm)
            for regs, v in vals(rm):
                em.setv(regs, v)
            em.run(E.order)
            return em.getv(out(rm))
        return ed._with_globals(go)

    edge = [0, 1, 18, 19, Q - 1, Q, Q + 1, B - 1, B, B + 18, LIM - 1]
    for t in range(60):
        a = rng.choice(edge) if t < 30 else rng.randrange(LIM)
        b = rng.choice(edge) if t % 2 == 0 else rng.randrange(LIM)
        m = run(lambda rm: ed.pmul(rm, rm.X1, rm.Y1, rm.Z1), lambda rm: ((rm.X1, a), (rm.Y1, b)), lambda rm: rm.Z1)
        assert m < LIM and m % Q == a * b % Q                       # no Montgomery factor: the plain product
        q2 = run(lambda rm: ed.psqr(rm, rm.X1, rm.Z1), lambda rm: ((rm.X1, a),), lambda rm: rm.Z1)
        assert q2 < LIM and q2 % Q == a * a % Q
        s_ = run(lambda rm: ed.add_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.A), lambda rm: ((rm.X1, a), (rm.Y1, b)), lambda rm: rm.Z1)
        assert s_ < B + 58 and (s_ - (a + b)) % Q == 0
        d = run(lambda rm: ed.sub_lz(rm, rm.X1, rm.Y1, rm.Z1, rm.A), lambda rm: ((rm.X1, a), (rm.Y1, b)), lambda rm: rm.Z1)
        assert d < B + 58 and (d - (a - b)) % Q == 0


def test_ed_projective_scaling_makes_montgomery_inputs_plain():
    """Why the loop needs no domain conversion: extended coordinates scaled by any non-zero factor are the same point, so the Montgomery-form
    limbs (X R, Y R, Z R, T R) read as plain elements -- and cached entries built from them -- give the right point through the plain-product
    bodies, and the plain result is again a valid representative.  One doubling + one addition on the emulator, inputs scaled by R."""
    import random
    rng = random.Random(31)
    Q, R = ed.Q, ed.R

    def go():
        rm = ed.RegMap()
        Ed_ = ed.Emitter(); Ed_.schedule(ed.seq_double(rm, True))
        Ea_ = ed.Emitter(); Ea_.schedule(ed.seq_add(rm, True))
        for _ in range(6):
            P = ed.ed_mul_aff(ed.ED_B, rng.randrange(1, ed.L_ORD)); P2 = ed.ed_mul_aff(ed.ED_B, rng.randrange(1, ed.L_ORD))
            z, z2 = rng.randrange(1, Q), rng.randrange(1, Q)
            X, Y, Z, T = (P[0] * z % Q, P[1] * z % Q, z, P[0] * P[1] * z % Q)
            X2, Y2, Z2, T2 = (P2[0] * z2 % Q, P2[1] * z2 % Q, z2, P2[0] * P2[1] * z2 % Q)
            em = ed._emu(rm)
            for regs, v in ((rm.X1, X), (rm.Y1, Y), (rm.Z1, Z), (rm.T1, T)):
                em.setv(regs, v * R % Q)                                                   # arkworks limbs, read as plain elements
            em.run(Ed_.order)
            cached = ((Y2 + X2) * R % Q, (Y2 - X2) * R % Q, 2 * ed.D_ED * T2 * R % Q, 2 * Z2 * R % Q)   # what k_ed_smul_prep stores (Montgomery form)
            for regs, v in zip((rm.QP, rm.QM, rm.QT, rm.QZ), cached):
                em.setv(regs, v)
            em.run(Ea_.order)
            gx, gy, gz, gt = (em.getv(r_) % Q for r_ in (rm.X1, rm.Y1, rm.Z1, rm.T1))
            zi = pow(gz, -1, Q)
            want = ed.ed_add_aff(ed.ed_add_aff(P, P), P2)
            assert (gx * zi % Q, gy * zi % Q) == want and gt * gz % Q == gx * gy % Q
    ed._with_globals(go)


def test_committed_ed_header_is_current(tmp_path):
    out = tmp_path / "ed_asm_kernels.inc"
    ed.emit_header(str(out))
    assert out.read_text() == open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ed_asm_kernels.inc")).read(), "regenerate with: python tools/gen_ed_asm.py"


# ---- round 3: the Curve25519 streams on nine 29-bit limbs (tools/gen_ed29_asm.py -> csrc/ed29_asm_kernels.inc) ---------------------------
import gen_ed29_asm as ed29  # noqa: E402


def test_ed29_bodies_match_the_affine_edwards_law():
    """Doubling (with / without T), cached addition, Niels addition on unsaturated limbs in plain arithmetic mod 2^255 - 19, scheduled as emitted,
    against the affine law (identity, P + P, P + (-P) included: the law is complete); the cached form, the packing (every stored word string
    below 2^256) and the in-place unpacking round-trip."""
    r = ed29.selftest(trials=36, seed=20260929)
    assert len(r[False][0].order) < 1000 and len(r[False][1].order) < 1100


def test_ed29_every_multiplication_at_its_operand_bounds():
    assert ed29.selftest_extremes() >= 14


def test_ed29_msm_member_path():
    """The MSM fold's per-member work on the emulator: affine (x, y) -> Niels form on the fly, negative-digit and past-the-end selection (the
    identity's Niels form), Niels addition -- against the affine law, identity / same point / opposite point as members included."""
    assert ed29.selftest_member(trials=36, seed=5)


def test_committed_ed29_header_is_current(tmp_path):
    out = tmp_path / "ed29_asm_kernels.inc"
    ed29.emit_header(str(out))
    committed = open(os.path.join(ROOT, "ark-mpc_amd", "csrc", "ed29_asm_kernels.inc")).read()
    assert out.read_text() == committed, "regenerate with: python tools/gen_ed29_asm.py"
