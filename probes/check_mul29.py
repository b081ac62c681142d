#!/usr/bin/env python3
"""Checks the one product probes/mulrate29 prints (x * x / 2^261 mod q in 29-bit limbs) against Python integers, and its constants.
usage: probes/mulrate29 | python probes/check_mul29.py"""
import json, sys
q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
assert (-pow(q, -1, 1 << 29)) % (1 << 29) == 0x04866389
for line in sys.stdin:
    d = json.loads(line)
    if d.get("check") == "mul29":
        x = sum(v << (29 * i) for i, v in enumerate(d["x"]))
        got = sum(v << (29 * i) for i, v in enumerate(d["x_times_x"]))
        want = x * x * pow(1 << 261, -1, q) % q
        assert got % q == want and got < 2 * q, (hex(got), hex(want))
        print("mul29 ok: x*x/2^261 mod q matches Python (result in [0, 2q))")
    else:
        if d.get("ok") is False:
            raise SystemExit("hand-scheduled block differs from the compiled one")
        print(line.strip())
