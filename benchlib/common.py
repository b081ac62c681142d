"""Shared pieces of the bench legs: constants of SURVEY.md section 8(d), the seeded synthetic workload (generated on the GPU through the
engine's own ops), the pre-bound launches of a step, the relational result check and small timing helpers."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "authenticated Beaver mul-gates/sec over BN254 Fr, batch 2^20, at 1/2/4/8 GPUs"
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_GATE = 1024       # two-party gate, SURVEY.md section 8(d)
# apportioning of the 512 B / party-gate of SURVEY 8(d) between the two kernels of a party:
ALG_BYTES_K1 = 192              # read x,y shares+MACs 128, write own d||e 64
ALG_BYTES_K3 = 320              # read a,b,c shares+MACs 192, read peer d||e 64, write result 64
FID = 0                         # BN254 Fr


def rand_field_elems(eng, n, gen):
    """n uniformly random BN254 Fr elements in Montgomery form (int64 tensor of 4n limbs), generated on the GPU."""
    raw = torch.randint(-(2**63), 2**63 - 1, (4 * n,), dtype=torch.int64, device="cuda", generator=gen)
    out = torch.empty_like(raw)
    eng.scalar_from_canonical(n, raw, out)   # reduces mod p, then to Montgomery form
    return out


def additive_split(eng, n, v, gen):
    s0 = rand_field_elems(eng, n, gen)
    s1 = torch.empty_like(s0)
    eng.scalar_sub(n, v, s0, s1)
    return s0, s1


def make_shares(eng, n, v, key, gen, layout):
    """SPDZ-share the vector v under MAC key `key` (both Montgomery limb tensors) -> per-party share buffers."""
    mac = torch.empty_like(v)
    eng.scalar_mul(n, v, key.repeat(n), mac)
    s0, s1 = additive_split(eng, n, v, gen)
    m0, m1 = additive_split(eng, n, mac, gen)
    if layout == "aos":   # [n][share(4) | mac(4)]
        p0 = torch.cat([s0.view(n, 4), m0.view(n, 4)], dim=1).contiguous().view(-1)
        p1 = torch.cat([s1.view(n, 4), m1.view(n, 4)], dim=1).contiguous().view(-1)
    else:                 # [share column (4n) | mac column (4n)]
        p0 = torch.cat([s0, m0]).contiguous()
        p1 = torch.cat([s1, m1]).contiguous()
    return p0, p1


class Party:
    pass


def build_workload(eng, n, seed, layout, key_shares=None):
    """key_shares: the two parties' MAC key shares (numpy 4 x u64 each) when this batch is one RANGE of a larger one that shares the key
    (the single-process group); by default they are drawn from the seed."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    key_sh = [rand_field_elems(eng, 1, gen), rand_field_elems(eng, 1, gen)]
    if key_shares is not None:
        key_sh = [torch.from_numpy(np.ascontiguousarray(k).view(np.int64)).to(key_sh[0].device) for k in key_shares]
    key = torch.empty_like(key_sh[0])
    eng.scalar_add(1, key_sh[0], key_sh[1], key)
    x = rand_field_elems(eng, n, gen)
    y = rand_field_elems(eng, n, gen)
    a = rand_field_elems(eng, n, gen)
    b = rand_field_elems(eng, n, gen)
    c = torch.empty_like(a)
    eng.scalar_mul(n, a, b, c)
    parties = [Party(), Party()]
    for name, v in (("x", x), ("y", y), ("a", a), ("b", b), ("c", c)):
        p0, p1 = make_shares(eng, n, v, key, gen, layout)
        setattr(parties[0], name, p0)
        setattr(parties[1], name, p1)
    for pid, p in enumerate(parties):
        p.id = pid
        p.key = key_sh[pid].cpu().numpy().view(np.uint64).copy()
        p.de = torch.empty(2 * n * 4, dtype=torch.int64, device="cuda")
        p.out = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    return parties, (x, y, key)


def prepare_step(eng, n, parties, layout, chunks=1, k3_order="01"):
    """Pre-bind the launches of a step (arguments marshalled once; buffers are fixed for the whole run).
    With chunks > 1 the batch is cut into gate ranges and each range runs K1(P0), K1(P1), K3(P0), K3(P1)."""
    S = lambda v: ("size", v)
    P = lambda t: t.data_ptr()
    calls = []
    m = n // chunks
    assert m * chunks == n
    for c in range(chunks):
        lo = c * m
        o8, o4 = lo * 64, lo * 32          # byte offsets of gate `lo` in AoS records / 32-byte columns
        de_off = c * 2 * m * 32             # each chunk owns a contiguous d||e block of 2m scalars
        for p in parties:
            if layout == "aos":
                calls.append(eng.prepare("beaver_mask", S(m), P(p.x) + o8, P(p.y) + o8, P(p.a) + o8, P(p.b) + o8, P(p.de) + de_off))
            else:
                calls.append(eng.prepare("beaver_mask_v", S(m), P(p.x) + o4, S(4), P(p.y) + o4, S(4), P(p.a) + o4, S(4), P(p.b) + o4, S(4),
                                         P(p.de) + de_off))
        pairs = ((parties[0], parties[1]), (parties[1], parties[0]))
        for p, peer in (pairs if k3_order == "01" else pairs[::-1]):   # the "network" = reading the peer's d||e
            if layout == "aos":
                calls.append(eng.prepare("beaver_finish_fused", S(m), ("int", p.id), ("key", p.key), P(p.de) + de_off, P(peer.de) + de_off,
                                         P(p.a) + o8, P(p.b) + o8, P(p.c) + o8, P(p.out) + o8))
            else:
                col = 4 * n * 8  # byte offset of the mac column
                calls.append(eng.prepare("beaver_finish_fused_v", S(m), ("int", p.id), ("key", p.key), P(p.de) + de_off, P(peer.de) + de_off,
                                         P(p.a) + o4, P(p.a) + col + o4, S(4), P(p.b) + o4, P(p.b) + col + o4, S(4),
                                         P(p.c) + o4, P(p.c) + col + o4, S(4), P(p.out) + o4, P(p.out) + col + o4, S(4)))
    return calls


def step(calls, eng=None, slot_base=None):
    """One step = the pre-bound launches in order.  With slot_base set, each launch gets a kernel-timer slot: HIP events bound
    to the kernel's own dispatch (hipExtLaunchKernelGGL), so its duration excludes the dispatch gap."""
    if slot_base is None:
        for c in calls:
            c()
        return
    for j, c in enumerate(calls):
        eng.kernel_timer_arm(slot_base + j)
        c()


def check_results(eng, n, parties, truth, layout):
    """open(batch_mul(x, y)) == x*y and the MAC relation holds (reference test_batch_mul, :1571-1594),
    using only engine ops; the bit-exact comparison with the oracle is tests/ and smoke()."""
    x, y, key = truth
    p0, p1 = parties
    if layout == "aos":
        s0, m0 = p0.out.view(n, 8)[:, :4].contiguous().view(-1), p0.out.view(n, 8)[:, 4:].contiguous().view(-1)
        s1, m1 = p1.out.view(n, 8)[:, :4].contiguous().view(-1), p1.out.view(n, 8)[:, 4:].contiguous().view(-1)
    else:
        s0, m0, s1, m1 = p0.out[:4 * n], p0.out[4 * n:], p1.out[:4 * n], p1.out[4 * n:]
    prod = torch.empty_like(x); eng.scalar_mul(n, x, y, prod)
    opened = torch.empty_like(x); eng.scalar_add(n, s0, s1, opened)
    mac = torch.empty_like(x); eng.scalar_add(n, m0, m1, mac)
    kprod = torch.empty_like(x); eng.scalar_mul(n, prod, key.repeat(n), kprod)
    torch.cuda.synchronize()
    return bool(torch.equal(opened, prod)) and bool(torch.equal(mac, kprod))


def timed_events(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps           # ms


E2E_UP_BYTES, E2E_DOWN_BYTES = 384, 128     # per party-gate over the host link: x, y, a, b, c records + the peer's d||e up; own d||e + result record down


def pcie_calibration(mib=256):
    """What the host link of this box gives plain pinned copies (the ceiling of the streaming path): one direction, and both at once."""
    m = mib << 20
    dev, dev2 = torch.empty(m, dtype=torch.uint8, device="cuda"), torch.empty(m, dtype=torch.uint8, device="cuda")
    h, h2 = torch.empty(m, dtype=torch.uint8).pin_memory(), torch.empty(m, dtype=torch.uint8).pin_memory()
    h.fill_(1); h2.fill_(2)
    s2 = torch.cuda.Stream()

    def both():
        dev.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(dev2, non_blocking=True)

    out = {}
    for name, fn, vol in (("h2d", lambda: dev.copy_(h, non_blocking=True), m), ("d2h", lambda: h2.copy_(dev2, non_blocking=True), m), ("both", both, 2 * m)):
        fn(); torch.cuda.synchronize()
        best = 0.0
        for _ in range(6):                               # best of six batches of four copies (the denominator of frac_of_measured_pcie: a low reading would flatter the path)
            t0 = time.perf_counter()
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            best = max(best, vol / ((time.perf_counter() - t0) / 4) / 1e9)
        out[name + "_GBps"] = best
    return out


def pinned_array(lib, nwords):
    """a numpy u64 array over an arkmpc_host_alloc block (pinned, recycled); returns (array, pointer)"""
    q = ctypes.c_void_p()
    if lib.arkmpc_host_alloc(ctypes.c_size_t(8 * nwords), ctypes.byref(q)) != 0:
        raise RuntimeError("arkmpc_host_alloc(%d bytes)" % (8 * nwords))
    return np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_uint64)), shape=(nwords,)), q



def load_oracle():
    """oracle/ark_oracle.c through tests/oracle_api.py: the CHECKER of the legs' results and the thing the cpu_baseline leg times -- never
    on a measured GPU path"""
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    import oracle_api
    return oracle_api.load()


def host_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def oracle_sample(n):
    """how many gates of a batch the oracle re-computes in a leg's check: every gate where the host has the threads for it (the GPU boxes: 256),
    a 2^16 sample elsewhere"""
    return n if host_cores() >= 16 else min(n, 1 << 16)
