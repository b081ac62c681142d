"""ctypes binding of include/arkmpc.h.  Thin by design: argument marshalling and error mapping only."""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FIELD_IDS = {"bn254_fr": 0, "bls12_381_fr": 1, "curve25519_fr": 2, "bn254_fq": 3, "curve25519_fq": 4}
FIELD_MODULI = {
    0: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    2: 2**252 + 27742317777372353535851937790883648493,
    3: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    4: 2**255 - 19,
}


class ArkMpcError(RuntimeError):
    pass


def lib_path():
    """the in-tree engine; ARKMPC_LIBRARY names another BUILD of the same engine (tools/crash_hunt.sh: the hazard build, lib/libarkmpc_hip_hazard.so)"""
    return os.environ.get("ARKMPC_LIBRARY") or os.path.join(_HERE, "lib", "libarkmpc_hip.so")


def load_library():
    """Load the HIP engine.  Raises (never falls back) when the library has not been built."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ArkMpcError(
                "HIP engine not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
        _LIB = ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        _LIB.arkmpc_last_error.restype = ctypes.c_char_p
        _LIB.arkmpc_version.restype = ctypes.c_char_p
    return _LIB


def declared_symbols(header=None):
    """Every function name declared in include/arkmpc.h."""
    header = header or os.path.join(_HERE, "..", "include", "arkmpc.h")
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(arkmpc_[a-z0-9_]+)\s*\(", txt)))


def exported_symbols():
    lib = load_library()
    return [s for s in declared_symbols() if hasattr(lib, s)]


def _ptr(x):
    """Device pointer (int), torch tensor (data_ptr) or numpy array (host pointer) -> c_void_p."""
    if x is None:
        return ctypes.c_void_p(0)
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ArkMpcError("numpy buffer must be C-contiguous")
        return ctypes.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    raise ArkMpcError("unsupported buffer type %r" % type(x))


def _key(k):
    a = np.ascontiguousarray(k, dtype=np.uint64).reshape(4)
    return a, ctypes.c_void_p(a.ctypes.data)


class Engine:
    """One arkmpc context = one (field, GPU).  Buffers are device pointers / torch tensors, or numpy
    arrays when constructed with host_buffers=True."""

    def __init__(self, field, device=0, host_buffers=False, stream=None):
        self.lib = load_library()
        self.field_id = FIELD_IDS[field] if isinstance(field, str) else int(field)
        self.host_buffers = bool(host_buffers)
        h = ctypes.c_void_p()
        rc = self.lib.arkmpc_ctx_create(self.field_id, int(device), ctypes.byref(h))
        if rc != 0:
            raise ArkMpcError("arkmpc_ctx_create failed with status %d (no GPU => no engine; there is no CPU fallback)" % rc)
        self.h = h
        if host_buffers:
            self._ck(self.lib.arkmpc_ctx_set_host_buffers(self.h, 1))
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.arkmpc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.arkmpc_last_error(self.h)
            raise ArkMpcError("arkmpc status %d: %s" % (rc, msg.decode() if msg else ""))

    def set_stream(self, stream_ptr):
        self._ck(self.lib.arkmpc_ctx_set_stream(self.h, ctypes.c_void_p(int(stream_ptr))))

    def kernel_timer_arm(self, slot):
        self._ck(self.lib.arkmpc_kernel_timer_arm(self.h, ctypes.c_int(int(slot))))

    def kernel_timer_ms(self, slot):
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.arkmpc_kernel_timer_ms(self.h, ctypes.c_int(int(slot)), ctypes.byref(ms)))
        return float(ms.value)

    def sync(self):
        self._ck(self.lib.arkmpc_sync(self.h))

    # ---- device-side ordering between two contexts' streams (arkmpc_event_*)
    def event_record(self):
        ev = ctypes.c_void_p()
        self._ck(self.lib.arkmpc_event_record(self.h, ctypes.byref(ev)))
        return ev

    def event_wait(self, ev):
        self._ck(self.lib.arkmpc_event_wait(self.h, ev))

    def event_destroy(self, ev):
        self.lib.arkmpc_event_destroy(ev)

    # ---- raw call helper: name, n, then pointers / scalars in ABI order
    def call(self, name, *args):
        fn = getattr(self.lib, "arkmpc_" + name)
        conv = [self.h]
        keep = []
        for a in args:
            if isinstance(a, tuple) and a[0] == "key":
                arr, p = _key(a[1])
                keep.append(arr)
                conv.append(p)
            elif isinstance(a, tuple) and a[0] == "size":
                conv.append(ctypes.c_size_t(int(a[1])))
            elif isinstance(a, tuple) and a[0] == "int":
                conv.append(ctypes.c_int(int(a[1])))
            elif isinstance(a, tuple) and a[0] == "ref":
                conv.append(a[1])
            else:
                conv.append(_ptr(a))
        self._ck(fn(*conv))

    def prepare(self, name, *args):
        """Pre-convert the arguments of one entry point and return a zero-argument callable that replays the call
        (for launch-bound loops: no per-call marshalling). Buffers must stay alive and in place."""
        fn = getattr(self.lib, "arkmpc_" + name)
        conv, keep = [self.h], []
        for a in args:
            if isinstance(a, tuple) and a[0] == "key":
                arr, p = _key(a[1]); keep.append(arr); conv.append(p)
            elif isinstance(a, tuple) and a[0] == "size":
                conv.append(ctypes.c_size_t(int(a[1])))
            elif isinstance(a, tuple) and a[0] == "int":
                conv.append(ctypes.c_int(int(a[1])))
            else:
                conv.append(_ptr(a))
        conv = tuple(conv)
        ck, h, lib = self._ck, self.h, self.lib

        def replay(_fn=fn, _conv=conv, _keep=keep):
            rc = _fn(*_conv)
            if rc != 0:
                ck(rc)
        return replay

    # ---- Scalar vectors
    def scalar_add(self, n, a, b, out): self.call("scalar_add", ("size", n), a, b, out)
    def scalar_sub(self, n, a, b, out): self.call("scalar_sub", ("size", n), a, b, out)
    def scalar_mul(self, n, a, b, out): self.call("scalar_mul", ("size", n), a, b, out)
    def scalar_neg(self, n, a, out): self.call("scalar_neg", ("size", n), a, out)
    def scalar_sum(self, n, a, out): self.call("scalar_sum", ("size", n), a, out)
    def scalar_product(self, n, a, out): self.call("scalar_product", ("size", n), a, out)
    def share_sum(self, n, a, out): self.call("share_sum", ("size", n), a, out)
    def scalar_prefix_product(self, n, a, out): self.call("scalar_prefix_product", ("size", n), a, out)
    def scalar_batch_inverse(self, n, a, out): self.call("scalar_batch_inverse", ("size", n), a, out)
    def scalar_from_canonical(self, n, a, out): self.call("scalar_from_canonical", ("size", n), a, out)
    def scalar_to_canonical(self, n, a, out): self.call("scalar_to_canonical", ("size", n), a, out)
    def scalar_to_bytes_be(self, n, a, out): self.call("scalar_to_bytes_be", ("size", n), a, out)

    # ---- ScalarShare vectors
    def share_add(self, n, a, b, out): self.call("share_add", ("size", n), a, b, out)
    def share_sub(self, n, a, b, out): self.call("share_sub", ("size", n), a, b, out)
    def share_neg(self, n, a, out): self.call("share_neg", ("size", n), a, out)
    def share_split(self, n, aos, s, m): self.call("share_split", ("size", n), aos, s, m)
    def share_join(self, n, s, m, aos): self.call("share_join", ("size", n), s, m, aos)
    def share_extract(self, n, a, out): self.call("share_extract", ("size", n), a, out)
    def share_add_public(self, n, party, key, a, pub, out): self.call("share_add_public", ("size", n), ("int", party), ("key", key), a, pub, out)
    def share_sub_public(self, n, party, key, a, pub, out): self.call("share_sub_public", ("size", n), ("int", party), ("key", key), a, pub, out)
    def share_mul_public(self, n, a, pub, out): self.call("share_mul_public", ("size", n), a, pub, out)

    # ---- Beaver multiplication
    def beaver_mask(self, n, x, y, a, b, out_de): self.call("beaver_mask", ("size", n), x, y, a, b, out_de)
    def open_combine(self, n, mine, peer, out): self.call("open_combine", ("size", n), mine, peer, out)
    def beaver_finish(self, n, party, key, d, e, a, b, c, out): self.call("beaver_finish", ("size", n), ("int", party), ("key", key), d, e, a, b, c, out)
    def beaver_finish_fused(self, n, party, key, my_de, peer_de, a, b, c, out):
        self.call("beaver_finish_fused", ("size", n), ("int", party), ("key", key), my_de, peer_de, a, b, c, out)
    def beaver_mask_v(self, n, xs, xst, ys, yst, as_, ast, bs, bst, out_de):
        self.call("beaver_mask_v", ("size", n), xs, ("size", xst), ys, ("size", yst), as_, ("size", ast), bs, ("size", bst), out_de)
    def beaver_finish_fused_v(self, n, party, key, my_de, peer_de, a_s, a_m, ast, b_s, b_m, bst, c_s, c_m, cst, o_s, o_m, ost):
        self.call("beaver_finish_fused_v", ("size", n), ("int", party), ("key", key), my_de, peer_de, a_s, a_m, ("size", ast),
                  b_s, b_m, ("size", bst), c_s, c_m, ("size", cst), o_s, o_m, ("size", ost))

    def beaver_mask_to(self, n, xs, xst, ys, yst, as_, ast, bs, bst, out_d, out_e):
        self.call("beaver_mask_to", ("size", n), xs, ("size", xst), ys, ("size", yst), as_, ("size", ast), bs, ("size", bst), out_d, out_e)
    def beaver_mask_dup(self, n, xs, xst, ys, yst, as_, ast, bs, bst, out_de, out_de_msg):
        self.call("beaver_mask_dup", ("size", n), xs, ("size", xst), ys, ("size", yst), as_, ("size", ast), bs, ("size", bst), out_de, out_de_msg)
    def beaver_finish_fused_from(self, n, party, key, my_d, my_e, peer_d, peer_e, a_s, a_m, ast, b_s, b_m, bst, c_s, c_m, cst, o_s, o_m, ost):
        self.call("beaver_finish_fused_from", ("size", n), ("int", party), ("key", key), my_d, my_e, peer_d, peer_e, a_s, a_m, ("size", ast),
                  b_s, b_m, ("size", bst), c_s, c_m, ("size", cst), o_s, o_m, ("size", ost))
    def share_add_public_v(self, n, party, key, a_s, a_m, ast, pub, o_s, o_m, ost):
        self.call("share_add_public_v", ("size", n), ("int", party), ("key", key), a_s, a_m, ("size", ast), pub, o_s, o_m, ("size", ost))
    def share_sub_public_v(self, n, party, key, a_s, a_m, ast, pub, o_s, o_m, ost):
        self.call("share_sub_public_v", ("size", n), ("int", party), ("key", key), a_s, a_m, ("size", ast), pub, o_s, o_m, ("size", ost))
    def share_mul_public_v(self, n, a_s, a_m, ast, pub, o_s, o_m, ost):
        self.call("share_mul_public_v", ("size", n), a_s, a_m, ("size", ast), pub, o_s, o_m, ("size", ost))

    # ---- streaming host-to-host Beaver multiplication (arkmpc_hostmul_*): numpy arrays in, numpy arrays out
    def hostmul_begin(self, n, x, y, a, b, c, out_de):
        """Phase 1; returns the session handle.  The arrays must stay alive and unmodified until hostmul_finish / hostmul_abort."""
        s = ctypes.c_void_p()
        self.call("hostmul_begin", ("size", n), x, y, a, b, c, out_de, ("ref", ctypes.byref(s)))
        return s
    def hostmul_poll_de(self, s):
        g = ctypes.c_size_t(0)
        rc = self.lib.arkmpc_hostmul_poll_de(s, ctypes.byref(g))
        if rc != 0:
            self._ck(rc)
        return int(g.value)
    def hostmul_wait_de(self, s): self._ck(self.lib.arkmpc_hostmul_wait_de(s))
    def hostmul_finish(self, s, party, key, peer_de, out):
        arr, pk = _key(key)
        self._ck(self.lib.arkmpc_hostmul_finish(s, ctypes.c_int(int(party)), pk, _ptr(peer_de), _ptr(out)))
    def hostmul_abort(self, s): self._ck(self.lib.arkmpc_hostmul_abort(s))
    NO_PIN = 1
    def hostmul_begin_range(self, n, x, y, a, b, c, out_d, out_e, flags=0):
        """Phase 1 of a session over a RANGE of a larger batch (d and e halves addressed separately); every vector may be a numpy array
        (pageable / pinned host memory), a torch tensor or an int device pointer on this engine's GPU."""
        s = ctypes.c_void_p()
        self._ck(self.lib.arkmpc_hostmul_begin_range(self.h, ctypes.c_size_t(int(n)), _ptr(x), _ptr(y), _ptr(a), _ptr(b), _ptr(c), _ptr(out_d), _ptr(out_e),
                                                     ctypes.c_uint(int(flags)), ctypes.byref(s)))
        return s
    def hostmul_finish_async(self, s, party, key, peer_d, peer_e, out):
        arr, pk = _key(key)
        self._ck(self.lib.arkmpc_hostmul_finish_async(s, ctypes.c_int(int(party)), pk, _ptr(peer_d), _ptr(peer_e), _ptr(out)))
    def hostmul_end(self, s): self._ck(self.lib.arkmpc_hostmul_end(s))
    def stats(self):
        """arkmpc_ctx_get_stats as a dict"""
        st = (ctypes.c_uint64 * 9)()
        self._ck(self.lib.arkmpc_ctx_get_stats(self.h, st))
        return {"hostmul_zero_copy_phases": (int(st[0]), int(st[1])), "hostmul_copy_phases": (int(st[2]), int(st[3])),
                "hostmul_device_bytes_last": int(st[4]), "hostmul_device_bytes_peak": int(st[5]),
                "batch_async_imports": int(st[6]), "batch_blocking_imports": int(st[7]), "zc_refused_reused_address": int(st[8])}
    # ---- device-batch handles (arkmpc_batch_*): the subset the asynchronous import needs
    SCALAR, SCALAR_SHARE = 0, 1
    AOS, SPLIT = 0, 1
    def batch_from_host(self, kind, layout, n, host_records, asynchronous=False):
        b = ctypes.c_void_p()
        fn = self.lib.arkmpc_batch_from_host_async if asynchronous else self.lib.arkmpc_batch_from_host
        self._ck(fn(self.h, ctypes.c_int(int(kind)), ctypes.c_int(int(layout)), ctypes.c_size_t(int(n)), _ptr(host_records), ctypes.byref(b)))
        return b
    def batch_acquire(self, b): self._ck(self.lib.arkmpc_batch_acquire(self.h, b))
    def batch_host_release(self, b): self._ck(self.lib.arkmpc_batch_host_release(self.h, b))
    def batch_destroy(self, b): self._ck(self.lib.arkmpc_batch_destroy(self.h, b))
    def batch_to_host(self, b, out): self._ck(self.lib.arkmpc_batch_to_host(self.h, b, _ptr(out)))
    def batch_ptrs(self, b):
        """(share / record pointer, MAC pointer or 0, stride in u64) of a batch"""
        self.lib.arkmpc_batch_data.restype = ctypes.c_void_p
        self.lib.arkmpc_batch_mac_data.restype = ctypes.c_void_p
        self.lib.arkmpc_batch_stride.restype = ctypes.c_size_t
        return int(self.lib.arkmpc_batch_data(b) or 0), int(self.lib.arkmpc_batch_mac_data(b) or 0), int(self.lib.arkmpc_batch_stride(b))
    @staticmethod
    def host_trim():
        return load_library().arkmpc_host_trim()
    def hostmul_begin_wire(self, n, x, y, a, b, c, result_id, out_frame):
        """Phase 1 with the payload as a wire frame (uint8 host array of capacity >= wire_frame_bound(2 n)); returns (session, frame length)."""
        s = ctypes.c_void_p()
        ln = ctypes.c_size_t(0)
        self._ck(self.lib.arkmpc_hostmul_begin_wire(self.h, ctypes.c_size_t(n), _ptr(x), _ptr(y), _ptr(a), _ptr(b), _ptr(c), ctypes.c_uint64(int(result_id)),
                                                    _ptr(out_frame), ctypes.c_size_t(out_frame.nbytes), ctypes.byref(ln), ctypes.byref(s)))
        return s, int(ln.value)
    def hostmul_finish_wire(self, s, party, key, peer_frame, peer_len, out):
        """Phase 2 from the peer's wire frame; returns the frame's result_id."""
        arr, pk = _key(key)
        rid = ctypes.c_uint64(0)
        self._ck(self.lib.arkmpc_hostmul_finish_wire(s, ctypes.c_int(int(party)), pk, _ptr(peer_frame), ctypes.c_size_t(int(peer_len)), _ptr(out), ctypes.byref(rid)))
        return int(rid.value)

    # ---- batch open + MAC check
    def mac_check_shares(self, n, key, opened, shares, out): self.call("mac_check_shares", ("size", n), ("key", key), opened, shares, out)
    def open_and_mac_check(self, n, key, shares, peer, out_opened, out_chk):
        self.call("open_and_mac_check", ("size", n), ("key", key), shares, peer, out_opened, out_chk)
    def open_and_mac_check_v(self, n, key, share_col, mac_col, stride, peer, out_opened, out_chk):
        self.call("open_and_mac_check_v", ("size", n), ("key", key), share_col, mac_col, ("size", stride), peer, out_opened, out_chk)
    def mac_verify(self, n, mine, peer):
        ok = ctypes.c_int(-1)
        self.call("mac_verify", ("size", n), mine, peer, ("ref", ctypes.byref(ok)))
        return bool(ok.value)
    def mac_verify_async(self, n, mine, peer): self.call("mac_verify_async", ("size", n), mine, peer)
    def mac_verify_result(self):
        ok = ctypes.c_int(-1)
        self._ck(self.lib.arkmpc_mac_verify_result(self.h, ctypes.byref(ok)))
        return bool(ok.value)
    def commit_sha3(self, n, values, blinder):
        out = np.zeros(4, dtype=np.uint64)
        self.call("commit_sha3", ("size", n), values, ("key", blinder), out)
        return out

    # ---- BN254 G1
    def g1_add(self, n, a, b, out): self.call("g1_add", ("size", n), a, b, out)
    def g1_sub(self, n, a, b, out): self.call("g1_sub", ("size", n), a, b, out)
    def g1_neg(self, n, a, out): self.call("g1_neg", ("size", n), a, out)
    def g1_scalar_mul(self, n, pts, scalars, out): self.call("g1_scalar_mul", ("size", n), pts, scalars, out)
    def g1_generator_mul(self, n, scalars, out): self.call("g1_generator_mul", ("size", n), scalars, out)
    def g1_to_affine(self, n, pts, out_xy, out_inf): self.call("g1_to_affine", ("size", n), pts, out_xy, out_inf)
    def g1_to_bytes(self, n, pts, out): self.call("g1_to_bytes", ("size", n), pts, out)
    def pointshare_add(self, n, a, b, out): self.call("pointshare_add", ("size", n), a, b, out)
    def pointshare_sub(self, n, a, b, out): self.call("pointshare_sub", ("size", n), a, b, out)
    def pointshare_neg(self, n, a, out): self.call("pointshare_neg", ("size", n), a, out)
    def pointshare_mul_public(self, n, shares, scalars, out): self.call("pointshare_mul_public", ("size", n), shares, scalars, out)
    def pointshare_add_public(self, n, party, key, shares, pub, out):
        self.call("pointshare_add_public", ("size", n), ("int", party), ("key", key), shares, pub, out)
    def pointshare_sub_public(self, n, party, key, shares, pub, out):
        self.call("pointshare_sub_public", ("size", n), ("int", party), ("key", key), shares, pub, out)
    def scalarshare_mul_generator(self, n, ss, out): self.call("scalarshare_mul_generator", ("size", n), ss, out)
    def scalarshare_mul_point(self, n, ss, pts, out): self.call("scalarshare_mul_point", ("size", n), ss, pts, out)
    def pointshare_extract(self, n, shares, out): self.call("pointshare_extract", ("size", n), shares, out)
    def point_mac_check_shares(self, n, key, opened, shares, out): self.call("point_mac_check_shares", ("size", n), ("key", key), opened, shares, out)
    def point_beaver_finish(self, n, party, key, d_open, eG_open, a, b, c, out, ed=False):
        self.call("edpoint_beaver_finish" if ed else "point_beaver_finish", ("size", n), ("int", party), ("key", key), d_open, eG_open, a, b, c, out)
    def fill(self, n, record, out):
        rec = np.ascontiguousarray(record, dtype=np.uint64)
        self._ck(self.lib.arkmpc_fill(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(rec.size), ctypes.c_void_p(rec.ctypes.data), _ptr(out)))
    def g1_from_bytes(self, n, data, out, out_ok): self.call("g1_from_bytes", ("size", n), data, out, out_ok)
    def g1_sum(self, n, pts, out): self.call("g1_sum", ("size", n), pts, out)
    def pointshare_sum(self, n, shares, out): self.call("pointshare_sum", ("size", n), shares, out)
    def g1_msm(self, n, pts, scalars, out): self.call("g1_msm", ("size", n), pts, scalars, out)
    def g1_msm_authenticated(self, n, pts, scalar_shares, out): self.call("g1_msm_authenticated", ("size", n), pts, scalar_shares, out)
    def commit_points_sha3(self, n, pts, blinders, out): self.call("commit_points_sha3", ("size", n), pts, blinders, out)
    def point_mac_verify(self, n, mine, peer, out_ok): self.call("point_mac_verify", ("size", n), mine, peer, out_ok)

    # ---- wire format (QuicTwoPartyNet frames: u64 length + serde_json text)
    def wire_frame_bound(self, n):
        out = ctypes.c_size_t(0); self.lib.arkmpc_wire_frame_bound(ctypes.c_size_t(int(n)), ctypes.byref(out)); return int(out.value)
    def wire_encode_scalar_batch(self, result_id, n, scalars, out_frame, out_cap):
        ln = ctypes.c_size_t(0)
        self._ck(self.lib.arkmpc_wire_encode_scalar_batch(self.h, ctypes.c_uint64(int(result_id)), ctypes.c_size_t(int(n)), _ptr(scalars), _ptr(out_frame),
                                                          ctypes.c_size_t(int(out_cap)), ctypes.byref(ln)))
        return int(ln.value)
    def wire_encode_bytes32(self, kind, result_id, n, records, out_frame, out_cap):
        ln = ctypes.c_size_t(0)
        self._ck(self.lib.arkmpc_wire_encode_bytes32(self.h, ctypes.c_int(int(kind)), ctypes.c_uint64(int(result_id)), ctypes.c_size_t(int(n)), _ptr(records),
                                                     _ptr(out_frame), ctypes.c_size_t(int(out_cap)), ctypes.byref(ln)))
        return int(ln.value)
    def wire_decode_scalar_batch(self, frame, frame_len, max_n, out_scalars):
        n = ctypes.c_size_t(0); rid = ctypes.c_uint64(0)
        self._ck(self.lib.arkmpc_wire_decode_scalar_batch(self.h, _ptr(frame), ctypes.c_size_t(int(frame_len)), ctypes.c_size_t(int(max_n)), _ptr(out_scalars),
                                                          ctypes.byref(n), ctypes.byref(rid)))
        return int(n.value), int(rid.value)
    def wire_decode_bytes32(self, frame, frame_len, max_n, out_records):
        n = ctypes.c_size_t(0); rid = ctypes.c_uint64(0); kind = ctypes.c_int(-1)
        self._ck(self.lib.arkmpc_wire_decode_bytes32(self.h, _ptr(frame), ctypes.c_size_t(int(frame_len)), ctypes.c_size_t(int(max_n)), _ptr(out_records),
                                                     ctypes.byref(n), ctypes.byref(rid), ctypes.byref(kind)))
        return int(n.value), int(rid.value), int(kind.value)

    # ---- Curve25519 (Edwards) points
    def ed_add(self, n, a, b, out): self.call("ed_add", ("size", n), a, b, out)
    def ed_from_bytes(self, n, data, out, out_ok): self.call("ed_from_bytes", ("size", n), data, out, out_ok)
    def ed_sub(self, n, a, b, out): self.call("ed_sub", ("size", n), a, b, out)
    def ed_neg(self, n, a, out): self.call("ed_neg", ("size", n), a, out)
    def ed_scalar_mul(self, n, pts, sc, out): self.call("ed_scalar_mul", ("size", n), pts, sc, out)
    def ed_generator_mul(self, n, sc, out): self.call("ed_generator_mul", ("size", n), sc, out)
    def ed_to_affine(self, n, pts, out): self.call("ed_to_affine", ("size", n), pts, out)
    def ed_to_bytes(self, n, pts, out): self.call("ed_to_bytes", ("size", n), pts, out)
    def edshare_add(self, n, a, b, out): self.call("edshare_add", ("size", n), a, b, out)
    def edshare_sub(self, n, a, b, out): self.call("edshare_sub", ("size", n), a, b, out)
    def edshare_neg(self, n, a, out): self.call("edshare_neg", ("size", n), a, out)
    def edshare_mul_public(self, n, sh, sc, out): self.call("edshare_mul_public", ("size", n), sh, sc, out)
    def edshare_add_public(self, n, party, key, sh, pub, out): self.call("edshare_add_public", ("size", n), ("int", party), ("key", key), sh, pub, out)
    def scalarshare_mul_ed_generator(self, n, ss, out): self.call("scalarshare_mul_ed_generator", ("size", n), ss, out)
    def edshare_sub_public(self, n, party, key, sh, pub, out): self.call("edshare_sub_public", ("size", n), ("int", party), ("key", key), sh, pub, out)
    def scalarshare_mul_ed_point(self, n, ss, pts, out): self.call("scalarshare_mul_ed_point", ("size", n), ss, pts, out)
    def edshare_extract(self, n, shares, out): self.call("edshare_extract", ("size", n), shares, out)
    def ed_mac_check_shares(self, n, key, opened, shares, out): self.call("ed_mac_check_shares", ("size", n), ("key", key), opened, shares, out)
    def ed_mac_verify(self, n, mine, peer, out_ok): self.call("ed_mac_verify", ("size", n), mine, peer, out_ok)
    def commit_ed_points_sha3(self, n, pts, blinders, out): self.call("commit_ed_points_sha3", ("size", n), pts, blinders, out)
    def ed_msm(self, n, pts, scalars, out): self.call("ed_msm", ("size", n), pts, scalars, out)
    def ed_msm_authenticated(self, n, pts, scalar_shares, out): self.call("ed_msm_authenticated", ("size", n), pts, scalar_shares, out)
    def ed_sum(self, n, pts, out): self.call("ed_sum", ("size", n), pts, out)
    def edshare_sum(self, n, shares, out): self.call("edshare_sum", ("size", n), shares, out)


class Group:
    """arkmpc_group: ONE process, G members (device ids may repeat).  Sharded vectors are lists of G device pointers (ints or torch
    tensors living on the member's device); see include/arkmpc.h for the segment convention."""

    AOS, SPLIT = 0, 1

    def __init__(self, field, device_ids):
        self.lib = load_library()
        self.lib.arkmpc_group_last_error.restype = ctypes.c_char_p
        self.lib.arkmpc_group_ctx.restype = ctypes.c_void_p
        self.field_id = FIELD_IDS[field] if isinstance(field, str) else int(field)
        ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = ctypes.c_void_p()
        rc = self.lib.arkmpc_group_create(self.field_id, len(device_ids), ids, ctypes.byref(h))
        if rc != 0:
            raise ArkMpcError("arkmpc_group_create failed with status %d (no GPU => no engine; there is no CPU fallback)" % rc)
        self.h = h
        self.G = len(device_ids)
        self.devices = [int(d) for d in device_ids]

    def close(self):
        if getattr(self, "h", None):
            self.lib.arkmpc_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.arkmpc_group_last_error(self.h)
            raise ArkMpcError("arkmpc group status %d: %s" % (rc, msg.decode() if msg else ""))

    def _sh(self, shards):
        if len(shards) != self.G:
            raise ArkMpcError("a sharded vector has one pointer per member")
        return (ctypes.c_void_p * self.G)(*[_ptr(x).value or 0 for x in shards])

    def shard_range(self, n, member):
        lo, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._ck(self.lib.arkmpc_group_shard_range(self.h, ctypes.c_size_t(int(n)), int(member), ctypes.byref(lo), ctypes.byref(cnt)))
        return int(lo.value), int(cnt.value)

    def peer_access(self, a, b):
        return bool(self.lib.arkmpc_group_peer_access(self.h, int(a), int(b)))

    def member_ctx(self, member):
        return ctypes.c_void_p(self.lib.arkmpc_group_ctx(self.h, int(member)))

    def sync(self):
        self._ck(self.lib.arkmpc_group_sync(self.h))

    def wait_group(self, producer):
        """member m of this group waits (on the device) for what member m of `producer` has submitted so far"""
        self._ck(self.lib.arkmpc_group_wait_group(self.h, producer.h))

    def malloc(self, n, segs, elem_words):
        out = (ctypes.c_void_p * self.G)()
        self._ck(self.lib.arkmpc_group_malloc(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(elem_words)), out))
        return [int(v or 0) for v in out]

    def free(self, shards):
        self._ck(self.lib.arkmpc_group_free(self.h, self._sh(shards)))

    def scatter_h2d(self, n, segs, ew, host, shards):
        self._ck(self.lib.arkmpc_group_scatter_h2d(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(ew)), _ptr(host), self._sh(shards)))

    def gather_d2h(self, n, segs, ew, shards, host):
        self._ck(self.lib.arkmpc_group_gather_d2h(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(ew)), self._sh(shards), _ptr(host)))

    def shares_from_host(self, layout, n, host_records, shards):
        self._ck(self.lib.arkmpc_group_shares_from_host(self.h, int(layout), ctypes.c_size_t(int(n)), _ptr(host_records), self._sh(shards)))

    def shares_to_host(self, layout, n, shards, host_records):
        self._ck(self.lib.arkmpc_group_shares_to_host(self.h, int(layout), ctypes.c_size_t(int(n)), self._sh(shards), _ptr(host_records)))

    def gather(self, n, segs, ew, shards, root, out_on_root):
        self._ck(self.lib.arkmpc_group_gather(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(ew)), self._sh(shards), int(root), _ptr(out_on_root)))

    def allgather(self, n, segs, ew, shards, outs):
        self._ck(self.lib.arkmpc_group_allgather(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(ew)), self._sh(shards), self._sh(outs)))

    def scatter(self, n, segs, ew, src_on_root, root, shards):
        self._ck(self.lib.arkmpc_group_scatter(self.h, ctypes.c_size_t(int(n)), ctypes.c_size_t(int(segs)), ctypes.c_size_t(int(ew)), _ptr(src_on_root), int(root), self._sh(shards)))

    def beaver_mask(self, layout, n, x, y, a, b, out_de):
        self._ck(self.lib.arkmpc_group_beaver_mask(self.h, int(layout), ctypes.c_size_t(int(n)), self._sh(x), self._sh(y), self._sh(a), self._sh(b), self._sh(out_de)))

    def beaver_mask_gathered(self, layout, n, x, y, a, b, root, out_de_on_root):
        self._ck(self.lib.arkmpc_group_beaver_mask_gathered(self.h, int(layout), ctypes.c_size_t(int(n)), self._sh(x), self._sh(y), self._sh(a), self._sh(b), int(root),
                                                            _ptr(out_de_on_root)))

    def beaver_finish_fused(self, layout, n, party, key, my_de, peer_de, a, b, c, out):
        arr, kp = _key(key)
        self._ck(self.lib.arkmpc_group_beaver_finish_fused(self.h, int(layout), ctypes.c_size_t(int(n)), int(party), kp, self._sh(my_de), self._sh(peer_de), self._sh(a),
                                                           self._sh(b), self._sh(c), self._sh(out)))

    # ---- streaming sessions over the group (arkmpc_group_hostmul_*): host numpy vectors in, host vectors out, one link per member
    def hostmul_begin(self, n, x, y, a, b, c, out_de):
        s = ctypes.c_void_p()
        self._ck(self.lib.arkmpc_group_hostmul_begin(self.h, ctypes.c_size_t(int(n)), _ptr(x), _ptr(y), _ptr(a), _ptr(b), _ptr(c), _ptr(out_de), ctypes.byref(s)))
        return s

    def hostmul_poll_de(self, s):
        g = ctypes.c_size_t(0)
        self._ck(self.lib.arkmpc_group_hostmul_poll_de(s, ctypes.byref(g)))
        return int(g.value)

    def hostmul_wait_de(self, s):
        self._ck(self.lib.arkmpc_group_hostmul_wait_de(s))

    def hostmul_finish(self, s, party, key, peer_de, out):
        arr, kp = _key(key)
        self._ck(self.lib.arkmpc_group_hostmul_finish(s, ctypes.c_int(int(party)), kp, _ptr(peer_de), _ptr(out)))

    def hostmul_abort(self, s):
        self._ck(self.lib.arkmpc_group_hostmul_abort(s))

    def member_stats(self, member):
        st = (ctypes.c_uint64 * 9)()
        self.lib.arkmpc_ctx_get_stats(self.member_ctx(member), st)
        return {"hostmul_zero_copy_phases": (int(st[0]), int(st[1])), "hostmul_copy_phases": (int(st[2]), int(st[3])),
                "hostmul_device_bytes_last": int(st[4]), "hostmul_device_bytes_peak": int(st[5]), "zc_refused_reused_address": int(st[8])}

    def prepare_beaver(self, layout, n, party, key, x, y, a, b, c, my_de, peer_de, out):
        """Pre-marshalled K1 and K2+K3 group calls (replayed in timing loops)."""
        arr, kp = _key(key)
        lib, h = self.lib, self.h
        a1 = (h, int(layout), ctypes.c_size_t(int(n)), self._sh(x), self._sh(y), self._sh(a), self._sh(b), self._sh(my_de))
        a3 = (h, int(layout), ctypes.c_size_t(int(n)), int(party), kp, self._sh(my_de), self._sh(peer_de), self._sh(a), self._sh(b), self._sh(c), self._sh(out))
        ck = self._ck

        def k1(_a=a1, _keep=arr):
            rc = lib.arkmpc_group_beaver_mask(*_a)
            if rc:
                ck(rc)

        def k3(_a=a3, _keep=arr):
            rc = lib.arkmpc_group_beaver_finish_fused(*_a)
            if rc:
                ck(rc)
        return k1, k3

    def share_extract(self, layout, n, shares, out_values):
        self._ck(self.lib.arkmpc_group_share_extract(self.h, int(layout), ctypes.c_size_t(int(n)), self._sh(shares), self._sh(out_values)))

    def open_and_mac_check(self, layout, n, key, shares, peer_values, out_opened, out_chk):
        arr, kp = _key(key)
        self._ck(self.lib.arkmpc_group_open_and_mac_check(self.h, int(layout), ctypes.c_size_t(int(n)), kp, self._sh(shares), self._sh(peer_values), self._sh(out_opened),
                                                          self._sh(out_chk)))

    def mac_verify(self, n, mine, peer):
        ok = ctypes.c_int(-1)
        self._ck(self.lib.arkmpc_group_mac_verify(self.h, ctypes.c_size_t(int(n)), self._sh(mine), self._sh(peer), ctypes.byref(ok)))
        return bool(ok.value)

    def commit_sha3(self, n, values, blinder):
        out = np.zeros(4, dtype=np.uint64)
        arr, kp = _key(blinder)
        self._ck(self.lib.arkmpc_group_commit_sha3(self.h, ctypes.c_size_t(int(n)), self._sh(values), kp, ctypes.c_void_p(out.ctypes.data)))
        return out

    def g1_msm(self, n, points, scalars):
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self.lib.arkmpc_group_g1_msm(self.h, ctypes.c_size_t(int(n)), self._sh(points), self._sh(scalars), ctypes.c_void_p(out.ctypes.data)))
        return out

    def ed_msm(self, n, points, scalars):
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.lib.arkmpc_group_ed_msm(self.h, ctypes.c_size_t(int(n)), self._sh(points), self._sh(scalars), ctypes.c_void_p(out.ctypes.data)))
        return out


def sha3_256(data: bytes) -> bytes:
    lib = load_library()
    out = (ctypes.c_uint8 * 32)()
    buf = (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    rc = lib.arkmpc_sha3_256(buf, ctypes.c_size_t(len(data)), out)
    if rc != 0:
        raise ArkMpcError("arkmpc_sha3_256 status %d" % rc)
    return bytes(out)
