"""ctypes binding of the CPU oracle (oracle/ark_oracle.h).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORA = None

u64p = ctypes.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "_build", "libark_oracle.so")


class Oracle:
    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.lib.ora_mac_verify.restype = ctypes.c_int
        self.lib.ora_g1_to_affine.restype = ctypes.c_int
        self.lib.ora_g1_from_bytes.restype = ctypes.c_int
        self.lib.ora_ed_from_bytes.restype = ctypes.c_int

    @staticmethod
    def _p(a):
        if a is None:
            return ctypes.c_void_p(0)
        assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "oracle buffers are contiguous numpy arrays"
        return ctypes.c_void_p(a.ctypes.data)

    def _call(self, name, *args):
        conv = []
        for a in args:
            if isinstance(a, (int, np.integer)):
                conv.append(ctypes.c_size_t(int(a)))
            else:
                conv.append(self._p(a))
        return getattr(self.lib, name)(*conv)

    # -- numpy-returning helpers (field id first, like the C API)
    def un(self, name, fid, n, a, out_words):
        out = np.zeros(n * out_words, dtype=np.uint64)
        self._call(name, fid, n, a, out)
        return out

    def bin(self, name, fid, n, a, b, out_words):
        out = np.zeros(n * out_words, dtype=np.uint64)
        self._call(name, fid, n, a, b, out)
        return out

    def from_canonical(self, fid, a): return self.un("ora_fp_batch_from_canonical", fid, len(a) // 4, a, 4)
    def to_canonical(self, fid, a): return self.un("ora_fp_batch_to_canonical", fid, len(a) // 4, a, 4)
    def scalar_add(self, fid, a, b): return self.bin("ora_scalar_batch_add", fid, len(a) // 4, a, b, 4)
    def scalar_sub(self, fid, a, b): return self.bin("ora_scalar_batch_sub", fid, len(a) // 4, a, b, 4)
    def scalar_mul(self, fid, a, b): return self.bin("ora_scalar_batch_mul", fid, len(a) // 4, a, b, 4)
    def scalar_neg(self, fid, a): return self.un("ora_scalar_batch_neg", fid, len(a) // 4, a, 4)
    def scalar_prefix_product(self, fid, a): return self.un("ora_scalar_prefix_product", fid, len(a) // 4, a, 4)
    def scalar_sum(self, fid, a): return self._red("ora_scalar_sum", fid, len(a) // 4, a, 4)
    def scalar_product(self, fid, a): return self._red("ora_scalar_product", fid, len(a) // 4, a, 4)
    def share_sum(self, fid, a): return self._red("ora_share_sum", fid, len(a) // 8, a, 8)
    def _red(self, name, fid, n, a, out_words):
        out = np.zeros(out_words, dtype=np.uint64)
        self._call(name, fid, n, np.ascontiguousarray(a, dtype=np.uint64), out)
        return out
    def scalar_batch_inverse(self, fid, a): return self.un("ora_scalar_batch_inverse", fid, len(a) // 4, a, 4)
    def share_add(self, fid, a, b): return self.bin("ora_share_batch_add", fid, len(a) // 8, a, b, 8)
    def share_sub(self, fid, a, b): return self.bin("ora_share_batch_sub", fid, len(a) // 8, a, b, 8)
    def share_neg(self, fid, a): return self.un("ora_share_batch_neg", fid, len(a) // 8, a, 8)

    def share_add_public(self, fid, party, key, a, pub, sub=False):
        n = len(a) // 8
        out = np.zeros(n * 8, dtype=np.uint64)
        self._call("ora_share_batch_sub_public" if sub else "ora_share_batch_add_public", fid, n, party, key, a, pub, out)
        return out

    def share_mul_public(self, fid, a, pub): return self.bin("ora_share_batch_mul_public", fid, len(a) // 8, a, pub, 8)

    def beaver_mask(self, fid, x, y, a, b):
        n = len(x) // 8
        out = np.zeros(2 * n * 4, dtype=np.uint64)
        self._call("ora_beaver_mask", fid, n, x, y, a, b, out)
        return out

    def open_combine(self, fid, mine, peer): return self.bin("ora_open_combine", fid, len(mine) // 4, mine, peer, 4)

    def beaver_finish(self, fid, party, key, d, e, a, b, c):
        n = len(a) // 8
        out = np.zeros(n * 8, dtype=np.uint64)
        self._call("ora_beaver_finish", fid, n, party, key, d, e, a, b, c, out)
        return out

    def batch_mul_9pass_local(self, fid, party, key, x, y, a, b, c, peer_de):
        n = len(a) // 8
        my_de = np.zeros(2 * n * 4, dtype=np.uint64)
        out = np.zeros(n * 8, dtype=np.uint64)
        scratch = np.zeros(64 * n, dtype=np.uint64)
        self._call("ora_batch_mul_9pass_local", fid, n, party, key, x, y, a, b, c, peer_de, my_de, out, scratch)
        return my_de, out

    def mac_check_shares(self, fid, key, opened, shares):
        n = len(opened) // 4
        out = np.zeros(n * 4, dtype=np.uint64)
        self._call("ora_mac_check_shares", fid, n, key, opened, shares, out)
        return out

    def mac_verify(self, fid, mine, peer):
        return bool(self._call("ora_mac_verify", fid, len(mine) // 4, mine, peer))

    def sha3_256(self, data: bytes):
        buf = np.frombuffer(data if data else b"\0", dtype=np.uint8).copy()
        out = np.zeros(32, dtype=np.uint8)
        self.lib.ora_sha3_256(self._p(buf), ctypes.c_size_t(len(data)), self._p(out))
        return out.tobytes()

    def commit_scalars(self, fid, values, blinder):
        out = np.zeros(4, dtype=np.uint64)
        self._call("ora_commit_scalars", fid, len(values) // 4, values, blinder, out)
        return out

    def commit_bytes(self, fid, data: bytes, blinder):
        buf = np.frombuffer(data, dtype=np.uint8).copy()
        out = np.zeros(4, dtype=np.uint64)
        self.lib.ora_commit_bytes(ctypes.c_int(fid), self._p(buf), ctypes.c_size_t(len(data)), self._p(blinder), self._p(out))
        return out

    def to_bytes_be(self, fid, a):
        n = len(a) // 4
        out = np.zeros(32 * n, dtype=np.uint8)
        f = ctypes.c_void_p(self.lib.ora_get_field(ctypes.c_int(fid)))
        for i in range(n):
            self.lib.ora_fp_to_bytes_be(f, self._p(a[4 * i:4 * i + 4].copy()), ctypes.c_void_p(out.ctypes.data + 32 * i))
        return out

    # -- curve
    def g1_generator(self):
        out = np.zeros(12, dtype=np.uint64); self.lib.ora_g1_generator(self._p(out)); return out
    def g1_identity(self):
        out = np.zeros(12, dtype=np.uint64); self.lib.ora_g1_identity(self._p(out)); return out
    def g1_batch_add(self, a, b):
        n = len(a) // 12; out = np.zeros(12 * n, dtype=np.uint64); self._call("ora_g1_batch_add", n, a, b, out); return out
    def g1_batch_scalar_mul(self, pts, scalars):
        n = len(pts) // 12; out = np.zeros(12 * n, dtype=np.uint64); self._call("ora_g1_batch_scalar_mul", n, pts, scalars, out); return out
    def g1_batch_to_affine(self, pts):
        n = len(pts) // 12; xy = np.zeros(8 * n, dtype=np.uint64); inf = np.zeros(n, dtype=np.uint8)
        self._call("ora_g1_batch_to_affine", n, pts, xy, inf); return xy, inf
    def g1_to_bytes(self, pts):
        n = len(pts) // 12; out = np.zeros(32 * n, dtype=np.uint8)
        for i in range(n):
            self.lib.ora_g1_to_bytes(self._p(pts[12 * i:12 * i + 12].copy()), ctypes.c_void_p(out.ctypes.data + 32 * i))
        return out
    def g1_from_bytes(self, data):
        n = len(data) // 32; out = np.zeros(12 * n, dtype=np.uint64); ok = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            ok[i] = self.lib.ora_g1_from_bytes(ctypes.c_void_p(data.ctypes.data + 32 * i), ctypes.c_void_p(out.ctypes.data + 96 * i))
        return out, ok
    def g1_sum(self, pts, stride=12, off=0):
        n = len(pts) // stride; out = np.zeros(12, dtype=np.uint64)
        self.lib.ora_g1_sum(ctypes.c_size_t(n), ctypes.c_void_p(pts.ctypes.data + 8 * off), ctypes.c_size_t(stride), self._p(out)); return out
    def g1_msm(self, pts, scalars, stride=4, off=0):
        n = len(pts) // 12; out = np.zeros(12, dtype=np.uint64)
        self.lib.ora_g1_msm(ctypes.c_size_t(n), self._p(pts), ctypes.c_void_p(scalars.ctypes.data + 8 * off), ctypes.c_size_t(stride), self._p(out)); return out
    def g1_msm_authenticated(self, pts, scalar_shares):
        n = len(pts) // 12; out = np.zeros(24, dtype=np.uint64); self._call("ora_g1_msm_authenticated", n, pts, scalar_shares, out); return out
    def g1_neg(self, pts):
        n = len(pts) // 12; out = np.zeros(12 * n, dtype=np.uint64)
        for i in range(n):
            self.lib.ora_g1_neg(self._p(pts[12 * i:12 * i + 12].copy()), ctypes.c_void_p(out.ctypes.data + 96 * i))
        return out
    def pointshare_add(self, a, b, sub=False):
        n = len(a) // 24; out = np.zeros(24 * n, dtype=np.uint64)
        self._call("ora_pointshare_batch_sub" if sub else "ora_pointshare_batch_add", n, a, b, out); return out
    def pointshare_neg(self, a):
        n = len(a) // 24; out = np.zeros(24 * n, dtype=np.uint64); self._call("ora_pointshare_batch_neg", n, a, out); return out
    def pointshare_mul_public(self, shares, scalars):
        n = len(shares) // 24; out = np.zeros(24 * n, dtype=np.uint64); self._call("ora_pointshare_batch_mul_public", n, shares, scalars, out); return out
    def pointshare_add_public(self, party, key, shares, pub):
        n = len(shares) // 24; out = np.zeros(24 * n, dtype=np.uint64)
        self._call("ora_pointshare_batch_add_public", n, party, key, shares, pub, out); return out
    def scalarshare_mul_generator(self, ss):
        n = len(ss) // 8; out = np.zeros(24 * n, dtype=np.uint64); self._call("ora_scalarshare_batch_mul_generator", n, ss, out); return out
    def scalarshare_mul_point(self, ss, pts):
        n = len(ss) // 8; out = np.zeros(24 * n, dtype=np.uint64); self._call("ora_scalarshare_batch_mul_point", n, ss, pts, out); return out

    # -- Curve25519 (Edwards)
    def ed_generator(self):
        out = np.zeros(16, dtype=np.uint64); self.lib.ora_ed_generator(self._p(out)); return out
    def ed_identity(self):
        out = np.zeros(16, dtype=np.uint64); self.lib.ora_ed_identity(self._p(out)); return out
    def ed_batch_add(self, a, b):
        n = len(a) // 16; out = np.zeros(16 * n, dtype=np.uint64); self._call("ora_ed_batch_add", n, a, b, out); return out
    def ed_batch_neg(self, a):
        n = len(a) // 16; out = np.zeros(16 * n, dtype=np.uint64); self._call("ora_ed_batch_neg", n, a, out); return out
    def ed_batch_scalar_mul(self, pts, scalars, n=None, p_div=1, s_div=1):
        n = n if n is not None else len(pts) // 16
        out = np.zeros(16 * n, dtype=np.uint64); self._call("ora_ed_batch_scalar_mul", n, pts, p_div, scalars, s_div, out); return out
    def ed_batch_to_affine(self, pts):
        n = len(pts) // 16; out = np.zeros(8 * n, dtype=np.uint64); self._call("ora_ed_batch_to_affine", n, pts, out); return out
    def ed_to_bytes(self, pts):
        n = len(pts) // 16; out = np.zeros(32 * n, dtype=np.uint8)
        for i in range(n):
            self.lib.ora_ed_to_bytes(self._p(pts[16 * i:16 * i + 16].copy()), ctypes.c_void_p(out.ctypes.data + 32 * i))
        return out
    def ed_from_bytes(self, data):
        n = len(data) // 32; out = np.zeros(16 * n, dtype=np.uint64); ok = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            ok[i] = self.lib.ora_ed_from_bytes(ctypes.c_void_p(data.ctypes.data + 32 * i), ctypes.c_void_p(out.ctypes.data + 128 * i))
        return out, ok
    def edshare_add_public(self, party, key, shares, pub):
        n = len(shares) // 32; out = np.zeros(32 * n, dtype=np.uint64)
        self._call("ora_edshare_batch_add_public", n, party, key, shares, pub, out); return out

    def edshare_sub_public(self, party, key, shares, pub):
        n = len(shares) // 32; out = np.zeros(32 * n, dtype=np.uint64)
        self._call("ora_edshare_batch_sub_public", n, party, key, shares, pub, out); return out
    def pointshare_sub_public(self, party, key, shares, pub):
        n = len(shares) // 24; out = np.zeros(24 * n, dtype=np.uint64)
        self._call("ora_pointshare_batch_sub_public", n, party, key, shares, pub, out); return out
    def point_mac_check_shares(self, key, opened, shares):
        n = len(opened) // 12; out = np.zeros(12 * n, dtype=np.uint64)
        self._call("ora_point_mac_check_shares", n, key, opened, shares, out); return out
    def ed_mac_check_shares(self, key, opened, shares):
        n = len(opened) // 16; out = np.zeros(16 * n, dtype=np.uint64)
        self._call("ora_ed_mac_check_shares", n, key, opened, shares, out); return out
    def ed_sum(self, pts, stride=16, off=0):
        n = len(pts) // stride; out = np.zeros(16, dtype=np.uint64)
        self.lib.ora_ed_sum(ctypes.c_size_t(n), ctypes.c_void_p(pts.ctypes.data + 8 * off), ctypes.c_size_t(stride), self._p(out)); return out
    def ed_msm(self, pts, scalars, stride=4):
        n = len(pts) // 16; out = np.zeros(16, dtype=np.uint64); self._call("ora_ed_msm", n, pts, scalars, stride, out); return out
    def ed_msm_authenticated(self, pts, scalar_shares):
        n = len(pts) // 16; out = np.zeros(32, dtype=np.uint64); self._call("ora_ed_msm_authenticated", n, pts, scalar_shares, out); return out
    def ed_is_identity_sum(self, a, b):
        self.lib.ora_ed_is_identity_sum.restype = ctypes.c_int
        return bool(self.lib.ora_ed_is_identity_sum(self._p(np.ascontiguousarray(a)), self._p(np.ascontiguousarray(b))))

    # -- range-parallel forms for the full-size parity tests
    @staticmethod
    def host_threads():
        return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def beaver_mask_mt(self, fid, x, y, a, b, nthreads=None):
        n = len(x) // 8
        out = np.zeros(2 * n * 4, dtype=np.uint64)
        self.lib.ora_beaver_mask_mt(ctypes.c_int(fid), ctypes.c_size_t(n), self._p(x), self._p(y), self._p(a), self._p(b), self._p(out),
                                    ctypes.c_int(nthreads or self.host_threads()))
        return out

    def batch_mul_9pass_mt(self, fid, party, key, x, y, a, b, c, peer_de, nthreads=None):
        n = len(a) // 8
        my_de = np.zeros(2 * n * 4, dtype=np.uint64)
        out = np.zeros(n * 8, dtype=np.uint64)
        rc = self.lib.ora_batch_mul_9pass_mt(ctypes.c_int(fid), ctypes.c_size_t(n), ctypes.c_int(party), self._p(key), self._p(x), self._p(y),
                                             self._p(a), self._p(b), self._p(c), self._p(peer_de), self._p(my_de), self._p(out),
                                             ctypes.c_int(nthreads or self.host_threads()))
        assert rc == 0
        return my_de, out

    def batch_mul_fused_mt(self, fid, party, key, x, y, a, b, c, peer_de, nthreads=None):
        """the fused single-pass form (authenticated_scalar.rs:799-843 per element): same words as the 9-pass"""
        n = len(a) // 8
        my_de = np.zeros(2 * n * 4, dtype=np.uint64)
        out = np.zeros(n * 8, dtype=np.uint64)
        rc = self.lib.ora_batch_mul_fused_mt(ctypes.c_int(fid), ctypes.c_size_t(n), ctypes.c_int(party), self._p(key), self._p(x), self._p(y),
                                             self._p(a), self._p(b), self._p(c), self._p(peer_de), self._p(my_de), self._p(out),
                                             ctypes.c_int(nthreads or self.host_threads()))
        assert rc == 0
        return my_de, out

    def open_and_mac_check_mt(self, fid, key, shares, peer, nthreads=None):
        n = len(shares) // 8
        opened = np.zeros(4 * n, dtype=np.uint64); chk = np.zeros(4 * n, dtype=np.uint64)
        self.lib.ora_open_and_mac_check_mt(ctypes.c_int(fid), ctypes.c_size_t(n), self._p(key), self._p(shares), self._p(peer), self._p(opened),
                                           self._p(chk), ctypes.c_int(nthreads or self.host_threads()))
        return opened, chk

    def pointshare_mul_public_mt(self, shares, scalars, nthreads=None):
        n = len(shares) // 24
        out = np.zeros(24 * n, dtype=np.uint64)
        self.lib.ora_pointshare_batch_mul_public_mt(ctypes.c_size_t(n), self._p(shares), self._p(scalars), self._p(out),
                                                    ctypes.c_int(nthreads or self.host_threads()))
        return out

    def ed_batch_scalar_mul_mt(self, pts, scalars, n, p_div=1, s_div=1, nthreads=None):
        out = np.zeros(16 * n, dtype=np.uint64)
        self.lib.ora_ed_batch_scalar_mul_mt(ctypes.c_size_t(n), self._p(pts), ctypes.c_size_t(p_div), self._p(scalars), ctypes.c_size_t(s_div), self._p(out),
                                            ctypes.c_int(nthreads or self.host_threads()))
        return out

    def ed_batch_to_affine_mt(self, pts, nthreads=None):
        n = len(pts) // 16
        out = np.zeros(8 * n, dtype=np.uint64)
        self.lib.ora_ed_batch_to_affine_mt(ctypes.c_size_t(n), self._p(pts), self._p(out), ctypes.c_int(nthreads or self.host_threads()))
        return out

    def g1_batch_to_affine_mt(self, pts, nthreads=None):
        n = len(pts) // 12
        xy = np.zeros(8 * n, dtype=np.uint64); inf = np.zeros(n, dtype=np.uint8)
        self.lib.ora_g1_batch_to_affine_mt(ctypes.c_size_t(n), self._p(pts), self._p(xy), self._p(inf), ctypes.c_int(nthreads or self.host_threads()))
        return xy, inf

    # -- PartyIDBeaverSource
    def dummy_mac_key_share(self, fid, party):
        out = np.zeros(4, dtype=np.uint64); self._call("ora_dummy_mac_key_share", fid, party, out); return out
    def dummy_triples(self, fid, party, n):
        a, b, c = (np.zeros(8 * n, dtype=np.uint64) for _ in range(3))
        self._call("ora_dummy_triples", fid, party, n, a, b, c); return a, b, c
    def dummy_local_input_masks(self, fid, party, n):
        m = np.zeros(4 * n, dtype=np.uint64); s = np.zeros(8 * n, dtype=np.uint64)
        self._call("ora_dummy_local_input_masks", fid, party, n, m, s); return m, s
    def dummy_counterparty_input_masks(self, fid, party, n):
        s = np.zeros(8 * n, dtype=np.uint64); self._call("ora_dummy_counterparty_input_masks", fid, party, n, s); return s


def load():
    global _ORA
    if _ORA is None:
        _ORA = Oracle(build())
        _ORA.lib.ora_get_field.restype = ctypes.c_void_p
    return _ORA
