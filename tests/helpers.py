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


class EngineAdapter:
    """Presents the HIP engine (host-buffer mode, through the C ABI) with the oracle binding's numpy-in /
    numpy-out method shapes, so a golden test body can run against either backend."""

    def __init__(self, pkg):
        self.pkg = pkg
        self.engines = {}

    def eng(self, fid):
        if fid not in self.engines:
            self.engines[fid] = self.pkg.Engine(fid, device=0, host_buffers=True)
        return self.engines[fid]

    def _o(self, n, w):
        return np.zeros(n * w, dtype=np.uint64)

    def scalar_add(self, fid, a, b): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_add(n, a, b, o); return o
    def scalar_sub(self, fid, a, b): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_sub(n, a, b, o); return o
    def scalar_mul(self, fid, a, b): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_mul(n, a, b, o); return o
    def scalar_neg(self, fid, a): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_neg(n, a, o); return o
    def from_canonical(self, fid, a): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_from_canonical(n, a, o); return o
    def to_canonical(self, fid, a): n = len(a) // 4; o = self._o(n, 4); self.eng(fid).scalar_to_canonical(n, a, o); return o
    def to_bytes_be(self, fid, a):
        n = len(a) // 4; o = np.zeros(32 * n, dtype=np.uint8); self.eng(fid).scalar_to_bytes_be(n, a, o); return o
    def beaver_mask(self, fid, x, y, a, b): n = len(x) // 8; o = self._o(2 * n, 4); self.eng(fid).beaver_mask(n, x, y, a, b, o); return o
    def open_combine(self, fid, mine, peer): n = len(mine) // 4; o = self._o(n, 4); self.eng(fid).open_combine(n, mine, peer, o); return o
    def beaver_finish(self, fid, party, key, d, e, a, b, c):
        n = len(a) // 8; o = self._o(n, 8); self.eng(fid).beaver_finish(n, party, key, d, e, a, b, c, o); return o
    def share_add_public(self, fid, party, key, a, pub, sub=False):
        n = len(a) // 8; o = self._o(n, 8)
        (self.eng(fid).share_sub_public if sub else self.eng(fid).share_add_public)(n, party, key, a, pub, o); return o
    def mac_check_shares(self, fid, key, opened, shares):
        n = len(opened) // 4; o = self._o(n, 4); self.eng(fid).mac_check_shares(n, key, opened, shares, o); return o
    def mac_verify(self, fid, mine, peer): return self.eng(fid).mac_verify(len(mine) // 4, mine, peer)
    def commit_scalars(self, fid, values, blinder): return self.eng(fid).commit_sha3(len(values) // 4, values, blinder)
    # curve (context field = BN254 Fr)
    def g1_batch_add(self, a, b): n = len(a) // 12; o = self._o(n, 12); self.eng(0).g1_add(n, a, b, o); return o
    def g1_neg(self, a): n = len(a) // 12; o = self._o(n, 12); self.eng(0).g1_neg(n, a, o); return o
    def g1_batch_scalar_mul(self, pts, sc): n = len(pts) // 12; o = self._o(n, 12); self.eng(0).g1_scalar_mul(n, pts, sc, o); return o
    def g1_batch_to_affine(self, pts):
        n = len(pts) // 12; xy = self._o(n, 8); inf = np.zeros(n, dtype=np.uint8); self.eng(0).g1_to_affine(n, pts, xy, inf); return xy, inf
    def g1_to_bytes(self, pts): n = len(pts) // 12; o = np.zeros(32 * n, dtype=np.uint8); self.eng(0).g1_to_bytes(n, pts, o); return o
    def pointshare_add(self, a, b, sub=False):
        n = len(a) // 24; o = self._o(n, 24); (self.eng(0).pointshare_sub if sub else self.eng(0).pointshare_add)(n, a, b, o); return o
    def pointshare_neg(self, a): n = len(a) // 24; o = self._o(n, 24); self.eng(0).pointshare_neg(n, a, o); return o
    def pointshare_mul_public(self, sh, sc): n = len(sh) // 24; o = self._o(n, 24); self.eng(0).pointshare_mul_public(n, sh, sc, o); return o
    def pointshare_add_public(self, party, key, sh, pub):
        n = len(sh) // 24; o = self._o(n, 24); self.eng(0).pointshare_add_public(n, party, key, sh, pub, o); return o
    def scalarshare_mul_generator(self, ss): n = len(ss) // 8; o = self._o(n, 24); self.eng(0).scalarshare_mul_generator(n, ss, o); return o
    def scalarshare_mul_point(self, ss, pts): n = len(ss) // 8; o = self._o(n, 24); self.eng(0).scalarshare_mul_point(n, ss, pts, o); return o
    def g1_msm(self, pts, sc): n = len(pts) // 12; o = self._o(1, 12); self.eng(0).g1_msm(n, pts, sc, o); return o
    def g1_msm_authenticated(self, pts, ss): n = len(pts) // 12; o = self._o(1, 24); self.eng(0).g1_msm_authenticated(n, pts, ss, o); return o
    # Curve25519 (context field = Curve25519 Fr): 16-word extended points
    def ed_batch_scalar_mul(self, pts, sc): n = len(pts) // 16; o = self._o(n, 16); self.eng(2).ed_scalar_mul(n, pts, sc, o); return o
    def ed_batch_add(self, a, b): n = len(a) // 16; o = self._o(n, 16); self.eng(2).ed_add(n, a, b, o); return o
    def ed_batch_neg(self, a): n = len(a) // 16; o = self._o(n, 16); self.eng(2).ed_neg(n, a, o); return o
    def ed_to_bytes(self, pts): n = len(pts) // 16; o = np.zeros(32 * n, dtype=np.uint8); self.eng(2).ed_to_bytes(n, pts, o); return o
    def ed_from_bytes(self, data):
        n = len(data) // 32; o = self._o(n, 16); ok = np.zeros(n, dtype=np.uint8); self.eng(2).ed_from_bytes(n, data, o, ok); return o, ok
    def ed_batch_to_affine(self, pts): n = len(pts) // 16; o = self._o(n, 8); self.eng(2).ed_to_affine(n, pts, o); return o
    def ed_generator_mul(self, sc): n = len(sc) // 4; o = self._o(n, 16); self.eng(2).ed_generator_mul(n, sc, o); return o
    def g1_from_bytes(self, data):
        n = len(data) // 32; o = self._o(n, 12); ok = np.zeros(n, dtype=np.uint8); self.eng(0).g1_from_bytes(n, data, o, ok); return o, ok


class FreshVA:
    """Pageable host memory at addresses that have NEVER been handed out before in this process: anonymous mappings placed one after the other in
    a private stretch of the address space (MAP_FIXED_NOREPLACE at a bump pointer), never reused unless the test says so (`remap`).  The engine
    lets kernels address a caller-registered vector in place only in the FIRST registered life of its addresses (csrc/arkmpc_internal.hpp
    PinRegistry::retired); numpy's own arrays recycle the addresses of earlier, freed arrays, so a test that asserts WHICH path ran takes its
    vectors from here."""
    BASE = 0x510000000000
    _next = [BASE]
    _libc = None
    PROT_RW, MAP_PRIVATE, MAP_ANONYMOUS, MAP_FIXED_NOREPLACE = 0x3, 0x02, 0x20, 0x100000

    @classmethod
    def _c(cls):
        if cls._libc is None:
            import ctypes
            lc = ctypes.CDLL(None, use_errno=True)
            lc.mmap.restype = ctypes.c_void_p
            lc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
            lc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
            cls._libc = lc
        return cls._libc

    @classmethod
    def _map(cls, addr, nbytes):
        import ctypes
        q = cls._c().mmap(addr, nbytes, cls.PROT_RW, cls.MAP_PRIVATE | cls.MAP_ANONYMOUS | cls.MAP_FIXED_NOREPLACE, -1, 0)
        if q is None or q == ctypes.c_void_p(-1).value or q != addr:
            raise OSError("mmap(MAP_FIXED_NOREPLACE) at %#x failed: errno %d" % (addr, ctypes.get_errno()))
        return q

    @classmethod
    def zeros(cls, nwords, dtype=np.uint64):
        import ctypes
        nbytes = max(4096, (nwords * np.dtype(dtype).itemsize + 4095) & ~4095)
        addr = cls._next[0]
        cls._next[0] += nbytes + (2 << 20)                        # a gap after every mapping: neighbours never share a page or a TLB fragment
        cls._map(addr, nbytes)
        ct = ctypes.c_uint64 if np.dtype(dtype) == np.uint64 else ctypes.c_uint8
        a = np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(ct)), shape=(nwords,))
        return a

    @classmethod
    def copy(cls, arr):
        a = cls.zeros(arr.size, arr.dtype)
        a[:] = arr
        return a

    @classmethod
    def release(cls, arr):
        """give the pages back (the ADDRESSES are still never handed out again)"""
        cls._c().munmap(arr.ctypes.data, max(4096, (arr.nbytes + 4095) & ~4095))

    @classmethod
    def remap(cls, arr):
        """free the mapping under `arr` and map NEW anonymous pages at the same address (what free + malloc of the same size does to a recycled
        address); the array then reads zeros.  The caller must have unregistered it first."""
        nbytes = max(4096, (arr.nbytes + 4095) & ~4095)
        addr = arr.ctypes.data
        if cls._c().munmap(addr, nbytes) != 0:
            raise OSError("munmap failed")
        cls._map(addr, nbytes)
        return arr
