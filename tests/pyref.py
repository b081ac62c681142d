"""Python big-int ground truth used to PIN the C oracle (tests only).

Field ops are exact arithmetic in Z/pZ: the canonical residue is unique, so any correct
implementation (arkworks included) must produce these values.  Curve ops are the affine group law
of BN254 G1 (y^2 = x^3 + 3 over Fq, generator (1, 2)).  SHA3 is hashlib.
"""
import hashlib

P = {
    0: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,  # BN254 Fr
    1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,  # BLS12-381 Fr
    2: 2**252 + 27742317777372353535851937790883648493,                       # Curve25519 Fr
    3: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,  # BN254 Fq
    4: 2**255 - 19,                                                          # Curve25519 Fq
}
R = 1 << 256
# arkworks' published constants (ark-bn254 / ark-bls12-381 / ark-curve25519 `FrConfig`)
PUBLISHED_R = {
    0: 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb,
    3: 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d,
    1: 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe,
    2: 0x0ffffffffffffffffffffffffffffffec6ef5bf4737dcf70d6ec31748d98951d,
}
PUBLISHED_INV = {0: 0xc2e1f593efffffff, 3: 0x87d20782e4866389, 1: 0xfffffffeffffffff, 2: 0xd2b51da312547e1b}


def to_mont(fid, v):
    return (v % P[fid]) * R % P[fid]


def from_mont(fid, m):
    return m * pow(R, -1, P[fid]) % P[fid]


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def unlimbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def to_bytes_be(fid, v):
    return (v % P[fid]).to_bytes(32, "big")


def commit(fid, values, blinder):
    """commitment.rs:71-86 on canonical integers."""
    h = hashlib.sha3_256()
    for v in values:
        h.update(to_bytes_be(fid, v))
    h.update(to_bytes_be(fid, blinder))
    return int.from_bytes(h.digest(), "big") % P[fid]


# ---- BN254 G1 affine arithmetic, None = identity
Q = P[3]
RORD = P[0]
G = (1, 2)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % Q)


def g1_mul(a, k):
    k %= RORD
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, a)
        a = g1_add(a, a)
        k >>= 1
    return acc


def g1_compress(a):
    """ark-serialize compressed SW encoding (x LE, bit7 = y > -y, bit6 = infinity)."""
    if a is None:
        b = bytearray(32)
        b[31] |= 0x40
        return bytes(b)
    x, y = a
    b = bytearray(x.to_bytes(32, "little"))
    if y > (Q - y) % Q:
        b[31] |= 0x80
    return bytes(b)


def g1_jacobian_mont(a, z=1):
    """Jacobian (X, Y, Z) Montgomery-form limbs of an affine point scaled by z; identity = (1,1,0)."""
    if a is None:
        return limbs(to_mont(3, 1)) + limbs(to_mont(3, 1)) + [0, 0, 0, 0]
    x, y = a
    return limbs(to_mont(3, x * z * z)) + limbs(to_mont(3, y * z * z * z)) + limbs(to_mont(3, z))


# ---- Curve25519 in twisted-Edwards form (-x^2 + y^2 = 1 + d x^2 y^2 over 2^255 - 19), affine points, identity (0, 1)
EQ = P[4]
EL = P[2]
ED_D = (-121665 * pow(121666, -1, EQ)) % EQ
ED_B = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)


def ed_add(a, b):
    x1, y1 = a
    x2, y2 = b
    k = ED_D * x1 * x2 * y1 * y2 % EQ
    x3 = (x1 * y2 + x2 * y1) * pow(1 + k, -1, EQ) % EQ
    y3 = (y1 * y2 + x1 * x2) * pow(1 - k, -1, EQ) % EQ
    return (x3, y3)


def ed_neg(a):
    return ((-a[0]) % EQ, a[1])


def ed_mul(a, k):
    k %= EL
    acc = (0, 1)
    while k:
        if k & 1:
            acc = ed_add(acc, a)
        a = ed_add(a, a)
        k >>= 1
    return acc


def ed_compress(a):
    """ark-serialize compressed twisted-Edwards encoding: y little-endian, bit 7 of the last byte set iff x > -x."""
    x, y = a
    b = bytearray(y.to_bytes(32, "little"))
    if x > (EQ - x) % EQ:
        b[31] |= 0x80
    return bytes(b)


def ed_decompress(b):
    """Inverse of ed_compress with arkworks' validation: None if the bytes are not a point of the prime-order subgroup."""
    y = int.from_bytes(b, "little") & ((1 << 255) - 1)
    flag = b[31] >> 7
    if y >= EQ:
        return None
    w = (y * y - 1) * pow(ED_D * y * y + 1, -1, EQ) % EQ
    x = pow(w, (EQ + 3) // 8, EQ)
    if (x * x - w) % EQ:
        if (x * x + w) % EQ:
            return None
        x = x * pow(2, (EQ - 1) // 4, EQ) % EQ
    if (x > (EQ - x) % EQ) != bool(flag):
        x = (EQ - x) % EQ
    pt = (x, y)
    acc = (0, 1)                                   # [l]P by plain double-and-add (ed_mul reduces its scalar mod l)
    for bit in bin(EL)[2:]:
        acc = ed_add(acc, acc)
        if bit == "1":
            acc = ed_add(acc, pt)
    if acc != (0, 1):
        return None
    return pt


def ed_extended_mont(a, z=1):
    """extended coordinates (X, Y, T, Z) = (x z, y z, x y z, z) as Montgomery limbs"""
    x, y = a
    return limbs(to_mont(4, x * z)) + limbs(to_mont(4, y * z)) + limbs(to_mont(4, x * y * z)) + limbs(to_mont(4, z))


# ---- wire format: QuicTwoPartyNet frames (network/quic.rs:303-306) ------------------------------------------------
def wire_frame(kind, result_id, records):
    """u64 LE length || serde_json::to_vec(NetworkOutbound{result_id, payload: kind(records)}) for 32-byte records.
    serde_json's compact writer = json.dumps with separators (",", ":"): no whitespace, fields in declaration order
    (result_id, payload; network.rs:36-42), externally tagged enum variant, byte strings as arrays of integers."""
    import json
    import struct
    body = json.dumps({"result_id": int(result_id), "payload": {kind: [list(r) for r in records]}}, separators=(",", ":")).encode()
    return struct.pack("<Q", len(body)) + body


def wire_scalar_records(fid, values):
    """Scalar::serialize = serialize_uncompressed: 32 canonical little-endian bytes (scalar.rs:186-192)"""
    return [int(v % P[fid]).to_bytes(32, "little") for v in values]
