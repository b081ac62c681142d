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
