"""Shared test helpers: seeded inputs in arkworks layout (numpy uint64 limb arrays)."""
import random

import numpy as np

import pyref

MASK64 = (1 << 64) - 1


def ints_to_limbs(vals):
    out = np.zeros(4 * len(vals), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[4 * i + k] = (v >> (64 * k)) & MASK64
    return out


def limbs_to_ints(arr, words=4):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, words)
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) for row in arr]


def mont_array(fid, vals):
    """canonical python ints -> Montgomery-form limb array (arkworks in-memory form)."""
    return ints_to_limbs([pyref.to_mont(fid, v) for v in vals])


def from_mont_array(fid, arr):
    return [pyref.from_mont(fid, m) for m in limbs_to_ints(arr)]


def edge_values(fid):
    p = pyref.P[fid]
    return [0, 1, 2, p - 1, p - 2, (1 << 255) % p, (1 << 192) - 1, (1 << 64), (p + 1) // 2, 0xFFFFFFFF, (1 << 128) + 12345]


def rand_values(fid, n, seed):
    rng = random.Random(seed)
    p = pyref.P[fid]
    return [rng.randrange(p) for _ in range(n)]


def mixed_values(fid, n, seed):
    ev = edge_values(fid)
    vals = (ev + rand_values(fid, max(0, n - len(ev)), seed))[:n]
    return vals


def interleave_shares(share_limbs, mac_limbs):
    """two n x 4 limb arrays -> n ScalarShares (AoS, 8 limbs each)."""
    s = np.asarray(share_limbs, dtype=np.uint64).reshape(-1, 4)
    m = np.asarray(mac_limbs, dtype=np.uint64).reshape(-1, 4)
    return np.ascontiguousarray(np.concatenate([s, m], axis=1).reshape(-1))


def split_secret(fid, vals, seed):
    """additive 2-party split of each value: returns (shares_p0, shares_p1) as python ints."""
    rng = random.Random(seed)
    p = pyref.P[fid]
    s0 = [rng.randrange(p) for _ in vals]
    s1 = [(v - a) % p for v, a in zip(vals, s0)]
    return s0, s1


def authenticated_shares(fid, vals, key, seed):
    """SPDZ sharing of vals under MAC key `key`: per-party AoS ScalarShare arrays (Montgomery limbs)."""
    p = pyref.P[fid]
    s0, s1 = split_secret(fid, vals, seed)
    m0, m1 = split_secret(fid, [(key * v) % p for v in vals], seed + 7919)
    a0 = interleave_shares(mont_array(fid, s0), mont_array(fid, m0))
    a1 = interleave_shares(mont_array(fid, s1), mont_array(fid, m1))
    return a0, a1


def splitmix64_stream(seed, count):
    """counter-based PRNG (vectorised splitmix64) -> uint64 array; used for the large synthetic workloads."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z
