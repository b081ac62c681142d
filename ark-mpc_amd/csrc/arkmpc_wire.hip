// arkmpc_wire.hip -- the wire format on either side of the hot path (first half):
// what QuicTwoPartyNet puts on the stream for the batches the engine produces and consumes.
//
// Reference: a message is NetworkOutbound{result_id: usize, payload: NetworkPayload} (online-phase/src/network.rs:33-60)
// written as  u64 little-endian length || serde_json::to_vec(&msg)  (network/quic.rs:303-306, read back at :226-251).
// serde_json's compact writer emits no whitespace, struct fields in declaration order and externally tagged enum
// variants, so a ScalarBatch message is exactly
//     {"result_id":<decimal>,"payload":{"ScalarBatch":[[b0,b1,...,b31],[...],...]}}
// where each element is the scalar's `serialize_uncompressed` bytes -- 32 bytes, canonical, little-endian
// (algebra/scalar/scalar.rs:186-192) -- written as a JSON array of decimal numbers (serde_json serialize_bytes).
// CurvePoint serialises the same way from `to_bytes()` = serialize_compressed (curve.rs:50-55, :103-108), variant
// "PointBatch".  The survey names this serde_json pass as where an opening's time goes on the QUIC path (section 8a row a5):
// ~115 text bytes per 32-byte value.
//
// Encoding (HBM-bound byte work): lengths per record -> exclusive scan -> each workgroup renders its 256 records into
// LDS and copies the contiguous text out with coalesced stores.  Decoding: count '[' per 4 KiB block -> scan -> scatter
// element start positions -> one thread per element parses and validates its <= 131 characters.  Scalars are then
// checked against the modulus (deserialize_uncompressed validates, scalar.rs:195-201) and converted to Montgomery form.
#include "arkmpc_internal.hpp"
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstring>

#define WIRE_TPB 256
#define WIRE_MAX_REC 131u   // ",[" + 32 x "255" + 31 x "," + "]"

namespace {

__device__ __forceinline__ u32 ndigits(u32 b) { return 1u + (b >= 10u) + (b >= 100u); }

// text length of record i (with its leading separator when i > 0)
__global__ void __launch_bounds__(WIRE_TPB) k_wire_lengths(size_t n, const unsigned char* recs, u64* lens) {
    const size_t i = (size_t)blockIdx.x * WIRE_TPB + threadIdx.x;
    if (i > n) return;
    if (i == n) { lens[i] = 0; return; }
    const uint4* q = reinterpret_cast<const uint4*>(recs + 32 * i);
    const uint4 a = q[0], b = q[1];
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 len = 2u + 31u + (i ? 1u : 0u);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int s = 0; s < 32; s += 8) len += ndigits((w[k] >> s) & 255u);
    lens[i] = len;
}

struct WireHeader {
    unsigned char text[80];
    u32 len;
};

// One workgroup renders records [blk*256, blk*256+256) into LDS, then streams the contiguous text to `out`.
__global__ void __launch_bounds__(WIRE_TPB) k_wire_render(size_t n, const unsigned char* recs, const u64* off, WireHeader hdr, unsigned char* out) {
    __shared__ unsigned char sm[WIRE_TPB * WIRE_MAX_REC + 16];
    const size_t first = (size_t)blockIdx.x * WIRE_TPB;
    const size_t i = first + threadIdx.x;
    const size_t last = (first + WIRE_TPB < n) ? first + WIRE_TPB : n;
    const u64 base = off[first], end = off[last];
    if (i < n) {
        unsigned char* p = sm + (off[i] - base);
        const uint4* q = reinterpret_cast<const uint4*>(recs + 32 * i);
        const uint4 a = q[0], b = q[1];
        const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (i) *p++ = ',';
        *p++ = '[';
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int s = 0; s < 32; s += 8) {
                const u32 v = (w[k] >> s) & 255u;
                if (k | s) *p++ = ',';
                if (v >= 100u) *p++ = (unsigned char)('0' + v / 100u);
                if (v >= 10u) *p++ = (unsigned char)('0' + (v / 10u) % 10u);
                *p++ = (unsigned char)('0' + v % 10u);
            }
        }
        *p++ = ']';
    }
    __syncthreads();
    unsigned char* dst = out + 8 + hdr.len + base;
    const u32 total = (u32)(end - base);
    // head bytes up to 16-byte alignment of dst, then 16-byte vector stores, then the tail
    const u32 mis = (u32)((16u - ((uintptr_t)dst & 15u)) & 15u);
    const u32 head = mis < total ? mis : total;
    if (threadIdx.x < head) dst[threadIdx.x] = sm[threadIdx.x];
    const u32 nvec = (total - head) / 16u;
    for (u32 v = threadIdx.x; v < nvec; v += WIRE_TPB) {
        const unsigned char* s = sm + head + 16u * v;     // LDS side is byte-addressed: assemble the vector
        u32 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (u32)s[4 * k] | ((u32)s[4 * k + 1] << 8) | ((u32)s[4 * k + 2] << 16) | ((u32)s[4 * k + 3] << 24);
        *reinterpret_cast<uint4*>(dst + head + 16u * v) = make_uint4(x[0], x[1], x[2], x[3]);
    }
    const u32 done = head + 16u * nvec;
    if (threadIdx.x < total - done) dst[done + threadIdx.x] = sm[done + threadIdx.x];
    if (blockIdx.x == 0) {
        // frame prefix, JSON header and trailer
        const u64 body = off[n];
        const u64 json_len = hdr.len + body + 3;
        if (threadIdx.x < 8) out[threadIdx.x] = (unsigned char)(json_len >> (8 * threadIdx.x));
        if (threadIdx.x < hdr.len) out[8 + threadIdx.x] = hdr.text[threadIdx.x];
        if (threadIdx.x < 3) out[8 + hdr.len + body + threadIdx.x] = (unsigned char)("]}}"[threadIdx.x]);
    }
}

// ---- decode ------------------------------------------------------------------------------------------------------
// The body is frame[lo, hi).  `frame` is 16-byte aligned, so every thread examines one aligned 16-byte vector and masks
// the bytes outside the body; positions are frame-relative.
// 16 bytes at the aligned offset `off`, never touching a byte at or beyond `limit` (the caller's frame length): the frame is a
// caller-owned buffer of exactly frame_len bytes, so the one vector that straddles its end is assembled byte by byte
__device__ __forceinline__ uint4 load16_clamped(const unsigned char* frame, size_t off, size_t limit) {
    if (off + 16 <= limit) return *reinterpret_cast<const uint4*>(frame + off);
    u32 w[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < 16 && off + k < limit; ++k) w[k >> 2] |= (u32)frame[off + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ u32 bracket_mask(const unsigned char* frame, size_t v, size_t lo, size_t hi) {
    if (16 * v + 16 <= lo || 16 * v >= hi) return 0;
    const uint4 q = load16_clamped(frame, 16 * v, hi + 3);          // the frame ends 3 bytes ("]}}") after the body
    const u32 w[4] = {q.x, q.y, q.z, q.w};
    u32 mask = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const size_t at = 16 * v + k;
        if (((w[k >> 2] >> (8 * (k & 3))) & 255u) == (u32)'[' && at >= lo && at < hi) mask |= 1u << k;
    }
    return mask;
}
__global__ void __launch_bounds__(WIRE_TPB) k_wire_count(const unsigned char* frame, size_t v0, size_t lo, size_t hi, u32* blk_cnt, u32 nblocks) {
    __shared__ u32 wsum[WIRE_TPB / 64];
    u32 c = __popc(bracket_mask(frame, v0 + (size_t)blockIdx.x * WIRE_TPB + threadIdx.x, lo, hi));
    for (int s = 32; s > 0; s >>= 1) c += __shfl_down(c, s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (blockIdx.x == 0) blk_cnt[nblocks] = 0;
    }
}
__global__ void __launch_bounds__(WIRE_TPB) k_wire_positions(const unsigned char* frame, size_t v0, size_t lo, size_t hi, const u32* blk_off, u64* pos,
                                                             size_t max_n) {
    __shared__ u32 cnt[WIRE_TPB];
    const size_t v = v0 + (size_t)blockIdx.x * WIRE_TPB + threadIdx.x;
    const u32 mask = bracket_mask(frame, v, lo, hi);
    const u32 c = __popc(mask);
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (u32 s = 1; s < WIRE_TPB; s <<= 1) {       // inclusive Hillis-Steele scan over 256 counts
        const u32 t = threadIdx.x >= s ? cnt[threadIdx.x - s] : 0;
        __syncthreads();
        cnt[threadIdx.x] += t;
        __syncthreads();
    }
    size_t idx = (size_t)blk_off[blockIdx.x] + cnt[threadIdx.x] - c;
    for (u32 k = 0; k < 16; ++k)
        if (mask & (1u << k)) { if (idx < max_n) pos[idx] = 16 * v + k; ++idx; }
}
// error bits
#define WIRE_E_SYNTAX 1
#define WIRE_E_RANGE 2
// One workgroup stages the text of its 256 elements in LDS (coalesced 16-byte loads), then one thread per element
// parses  '[' number (',' number){31} ']'  followed by ',' + the next element, or by the end of the body.
__global__ void __launch_bounds__(WIRE_TPB) k_wire_parse(size_t n, size_t lo, size_t hi, const unsigned char* frame, const u64* pos, unsigned char* recs,
                                                         int* err) {
    __shared__ uint4 smv[(WIRE_TPB * WIRE_MAX_REC + 48) / 16 + 1];
    unsigned char* sm = reinterpret_cast<unsigned char*>(smv);
    const size_t first = (size_t)blockIdx.x * WIRE_TPB;
    const size_t i = first + threadIdx.x;
    const size_t last = (first + WIRE_TPB < n) ? first + WIRE_TPB : n;
    const size_t span_lo = pos[first], span_hi = (last < n) ? pos[last] : hi;
    const size_t al = span_lo & ~(size_t)15;
    // an element longer than the grammar allows makes the span exceed the staging buffer: syntax error, no parse
    const bool fits = span_hi - al <= (size_t)WIRE_TPB * WIRE_MAX_REC + 16;
    if (!fits) { if (threadIdx.x == 0) atomicOr(err, WIRE_E_SYNTAX); return; }
    const u32 nvec = (u32)((span_hi - al + 15) / 16);
    for (u32 v = threadIdx.x; v < nvec; v += WIRE_TPB) smv[v] = load16_clamped(frame, al + 16 * (size_t)v, hi + 3);   // up to the end of the frame, not beyond
    __syncthreads();
    if (i >= n) return;
    const size_t end = span_hi - al;                  // LDS-relative end of this workgroup's text
    size_t p = pos[i] - al;
    int bad = 0;
    if (i == 0 && pos[0] != lo) bad |= WIRE_E_SYNTAX;
    ++p;                                              // past '['
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int s = 0; s < 32; s += 8) {
            u32 v = 0, nd = 0;
            unsigned char lead = 0;
            while (p < end && nd < 4) {
                const unsigned char ch = sm[p];
                if (ch < '0' || ch > '9') break;
                if (nd == 0) lead = ch;
                v = v * 10u + (u32)(ch - '0');
                ++nd; ++p;
            }
            if (nd == 0 || nd > 3 || (nd > 1 && lead == '0')) bad |= WIRE_E_SYNTAX;
            else if (v > 255u) bad |= WIRE_E_RANGE;
            w[k] |= (v & 255u) << s;
            const unsigned char sep = (k == 7 && s == 24) ? ']' : ',';
            if (p >= end || sm[p] != sep) bad |= WIRE_E_SYNTAX;
            ++p;
        }
    }
    if (i + 1 < n) {
        // ',' then the next element's '[' immediately (its position is known from pos[])
        if (pos[i + 1] - al != p + 1 || (p < end ? sm[p] != ',' : (al + p >= hi || frame[al + p] != ','))) bad |= WIRE_E_SYNTAX;
    } else if (al + p != hi) {
        bad |= WIRE_E_SYNTAX;
    }
    uint4* q = reinterpret_cast<uint4*>(recs + 32 * i);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    if (bad) atomicOr(err, bad);
}
// canonical little-endian -> Montgomery, rejecting values >= p (deserialize_uncompressed validates)
template <int F>
__global__ void __launch_bounds__(WIRE_TPB) k_wire_scalars(size_t n, const unsigned char* recs, u64* out, int* err) {
    const size_t i = (size_t)blockIdx.x * WIRE_TPB + threadIdx.x;
    if (i >= n) return;
    using P = FieldParams<F>;
    const Fe v = fe_load(reinterpret_cast<const u64*>(recs + 32 * i));
    u32 br = 0, bo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { (void)__builtin_subc(v.v[k], P::P(k), br, &bo); br = bo; }
    if (!br) { atomicOr(err, WIRE_E_RANGE); return; }          // v - p did not borrow: v >= p
    fe_store(out + 4 * i, fe_from_canonical<F>(v));
}
template <int F>
__global__ void __launch_bounds__(WIRE_TPB) k_wire_to_canonical(size_t n, const u64* in, unsigned char* recs) {
    const size_t i = (size_t)blockIdx.x * WIRE_TPB + threadIdx.x;
    if (i >= n) return;
    fe_store(reinterpret_cast<u64*>(recs + 32 * i), fe_to_canonical<F>(fe_load(in + 4 * i)));
}


// ---- serde's object semantics -------------------------------------------------------------------------------------------
// `NetworkOutbound` derives Deserialize (network.rs:33-60) and QuicTwoPartyNet reads it with serde_json::from_slice
// (network/quic.rs:233-251): the two fields may come in ANY order, unknown fields are skipped (whatever JSON value they hold), a known
// field given twice is an error ("duplicate field"), keys are compared after unescaping ("result_id" IS result_id), the payload is an
// externally tagged enum -- an object with exactly ONE key naming the variant --, and only whitespace may follow the closing brace.
// A frame the strict GPU parser rejects is therefore re-read here, on the host, by a small JSON scanner (RFC 8259: strings with escapes and
// valid UTF-8, no raw control characters, no lone surrogates; nesting limited to 128 like serde_json's recursion limit); if it is a
// NetworkOutbound message in any of those shapes it is rewritten into the canonical compact frame and handed to the GPU parser again.
struct JsonScan {
    const unsigned char* p; size_t n, i; std::string err;
    bool fail(const char* what) { if (err.empty()) err = what; return false; }
    void ws() { while (i < n && (p[i] == 0x20 || p[i] == 0x09 || p[i] == 0x0a || p[i] == 0x0d)) ++i; }
    static int hexv(unsigned char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1)); }
    bool hex4(u32* out) {
        if (i + 4 > n) return fail("malformed message: truncated \\u escape");
        u32 v = 0;
        for (int k = 0; k < 4; ++k) { const int h = hexv(p[i + k]); if (h < 0) return fail("malformed message: bad \\u escape"); v = v * 16 + (u32)h; }
        i += 4; *out = v; return true;
    }
    static void put_utf8(std::string& o, u32 c) {
        if (c < 0x80) o.push_back((char)c);
        else if (c < 0x800) { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 63))); }
        else if (c < 0x10000) { o.push_back((char)(0xE0 | (c >> 12))); o.push_back((char)(0x80 | ((c >> 6) & 63))); o.push_back((char)(0x80 | (c & 63))); }
        else { o.push_back((char)(0xF0 | (c >> 18))); o.push_back((char)(0x80 | ((c >> 12) & 63))); o.push_back((char)(0x80 | ((c >> 6) & 63))); o.push_back((char)(0x80 | (c & 63))); }
    }
    // at an opening quote: consumes the string, returns its unescaped value
    bool string(std::string* out) {
        if (i >= n || p[i] != '"') return fail("malformed message: expected a string");
        ++i;
        std::string o;
        while (true) {
            if (i >= n) return fail("malformed message: unterminated string");
            const unsigned char c = p[i];
            if (c == '"') { ++i; break; }
            if (c < 0x20) return fail("malformed message: control character in a string");
            if (c == '\\') {
                if (i + 1 >= n) return fail("malformed message: truncated escape");
                const unsigned char e = p[i + 1]; i += 2;
                switch (e) {
                    case '"': o.push_back('"'); break; case '\\': o.push_back('\\'); break; case '/': o.push_back('/'); break;
                    case 'b': o.push_back('\b'); break; case 'f': o.push_back('\f'); break; case 'n': o.push_back('\n'); break;
                    case 'r': o.push_back('\r'); break; case 't': o.push_back('\t'); break;
                    case 'u': {
                        u32 u; if (!hex4(&u)) return false;
                        if (u >= 0xDC00 && u <= 0xDFFF) return fail("malformed message: lone trailing surrogate");
                        if (u >= 0xD800 && u <= 0xDBFF) {
                            u32 lo;
                            if (i + 2 > n || p[i] != '\\' || p[i + 1] != 'u') return fail("malformed message: lone leading surrogate");
                            i += 2; if (!hex4(&lo)) return false;
                            if (lo < 0xDC00 || lo > 0xDFFF) return fail("malformed message: lone leading surrogate");
                            u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        put_utf8(o, u); break;
                    }
                    default: return fail("malformed message: bad escape");
                }
                continue;
            }
            if (c < 0x80) { o.push_back((char)c); ++i; continue; }
            // raw multi-byte UTF-8: validate (from_slice rejects invalid UTF-8 inside strings)
            int len = (c >= 0xC2 && c <= 0xDF) ? 2 : ((c >= 0xE0 && c <= 0xEF) ? 3 : ((c >= 0xF0 && c <= 0xF4) ? 4 : 0));
            if (!len || i + len > n) return fail("malformed message: invalid UTF-8");
            for (int k = 1; k < len; ++k) if ((p[i + k] & 0xC0) != 0x80) return fail("malformed message: invalid UTF-8");
            if ((c == 0xE0 && p[i + 1] < 0xA0) || (c == 0xED && p[i + 1] > 0x9F) || (c == 0xF0 && p[i + 1] < 0x90) || (c == 0xF4 && p[i + 1] > 0x8F))
                return fail("malformed message: invalid UTF-8");
            o.append((const char*)p + i, len); i += len;
        }
        if (out) *out = std::move(o);
        return true;
    }
    bool number() {                    // RFC 8259 number grammar
        if (i < n && p[i] == '-') ++i;
        if (i >= n) return fail("malformed message: number");
        if (p[i] == '0') ++i;
        else if (p[i] >= '1' && p[i] <= '9') { while (i < n && p[i] >= '0' && p[i] <= '9') ++i; }
        else return fail("malformed message: number");
        if (i < n && p[i] == '.') { ++i; size_t d = 0; while (i < n && p[i] >= '0' && p[i] <= '9') { ++i; ++d; } if (!d) return fail("malformed message: number"); }
        if (i < n && (p[i] == 'e' || p[i] == 'E')) {
            ++i; if (i < n && (p[i] == '+' || p[i] == '-')) ++i;
            size_t d = 0; while (i < n && p[i] >= '0' && p[i] <= '9') { ++i; ++d; } if (!d) return fail("malformed message: number");
        }
        return true;
    }
    bool literal(const char* lit) { const size_t l = strlen(lit); if (i + l > n || memcmp(p + i, lit, l) != 0) return fail("malformed message: literal"); i += l; return true; }
    // any JSON value, validated and skipped
    bool value(int depth) {
        if (depth > 128) return fail("malformed message: recursion limit exceeded");
        ws();
        if (i >= n) return fail("malformed message: unexpected end");
        const unsigned char c = p[i];
        if (c == '"') return string(nullptr);
        if (c == '{') {
            ++i; ws();
            if (i < n && p[i] == '}') { ++i; return true; }
            while (true) {
                ws(); if (!string(nullptr)) return false;
                ws(); if (i >= n || p[i] != ':') return fail("malformed message: expected ':'"); ++i;
                if (!value(depth + 1)) return false;
                ws(); if (i >= n) return fail("malformed message: unexpected end");
                if (p[i] == ',') { ++i; continue; }
                if (p[i] == '}') { ++i; return true; }
                return fail("malformed message: expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++i; ws();
            if (i < n && p[i] == ']') { ++i; return true; }
            while (true) {
                if (!value(depth + 1)) return false;
                ws(); if (i >= n) return fail("malformed message: unexpected end");
                if (p[i] == ',') { ++i; continue; }
                if (p[i] == ']') { ++i; return true; }
                return fail("malformed message: expected ',' or ']'");
            }
        }
        if (c == 't') return literal("true");
        if (c == 'f') return literal("false");
        if (c == 'n') return literal("null");
        return number();
    }
};
// appends in[a, b) without inter-token whitespace (the span holds no strings that matter: the GPU parser rejects any in a batch body)
static bool append_stripped(const unsigned char* in, size_t a, size_t b, std::vector<unsigned char>& out) {
    bool in_string = false;
    for (size_t i = a; i < b; ++i) {
        const unsigned char c = in[i];
        const bool ws = c == 0x20 || c == 0x09 || c == 0x0a || c == 0x0d;
        if (in_string && c == '\\' && i + 1 < b) { out.push_back(c); out.push_back(in[++i]); continue; }
        if (c == '"') in_string = !in_string;
        if (!ws || in_string) { out.push_back(c); continue; }
        size_t j = i;
        while (j < b && (in[j] == 0x20 || in[j] == 0x09 || in[j] == 0x0a || in[j] == 0x0d)) ++j;
        const bool digit_before = !out.empty() && out.back() >= '0' && out.back() <= '9';
        const bool digit_after = j < b && in[j] >= '0' && in[j] <= '9';
        if (digit_before && digit_after) return false;
        i = j - 1;
    }
    return true;
}
// in: a whole frame (8-byte prefix + JSON text).  out: the canonical compact frame of the same message, or err set.
// Both forms a serde-derived struct deserialises from are read (serde's derive implements visit_map AND visit_seq): the object
// {"result_id": .., "payload": ..} with the semantics listed in include/arkmpc.h, and the two-element sequence [result_id, payload] --
// fields in declaration order (network.rs:33-40), a missing or a third element is an error (invalid length / trailing characters).
static bool normalise_message(const std::vector<unsigned char>& in, std::vector<unsigned char>& out, std::string& err) {
    JsonScan sc{in.data(), in.size(), 8, {}};
    auto bad = [&](const char* what) { err = sc.err.empty() ? what : sc.err; return false; };
    size_t rid_a = 0, rid_b = 0, arr_a = 0, arr_b = 0;
    bool have_rid = false, have_payload = false;
    std::string variant;
    // the two fields' values, wherever they stand
    auto read_rid = [&]() -> bool {
        sc.ws(); rid_a = sc.i;
        if (sc.i >= sc.n || !((sc.p[sc.i] >= '0' && sc.p[sc.i] <= '9') || sc.p[sc.i] == '-')) return bad("malformed message: result_id");
        if (!sc.number()) return bad("malformed message: result_id");
        rid_b = sc.i;
        return true;
    };
    auto read_payload = [&]() -> bool {
        sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != '{') return bad("malformed message: payload");
        ++sc.i; sc.ws();
        if (!sc.string(&variant)) return bad("malformed message: payload variant");
        sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != ':') return bad("malformed message: payload");
        ++sc.i; sc.ws(); arr_a = sc.i;
        if (sc.i >= sc.n || sc.p[sc.i] != '[') return bad("malformed message: payload");
        if (!sc.value(1)) return bad("malformed message: payload");
        arr_b = sc.i; sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != '}') return bad("malformed message: payload is not a single-variant object");
        ++sc.i;
        return true;
    };
    sc.ws();
    if (sc.i < sc.n && sc.p[sc.i] == '[') {                  // the sequence form
        ++sc.i;
        if (!read_rid()) return false;
        have_rid = true;
        sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != ',') return bad("malformed message: a sequence needs two elements (result_id, payload)");
        ++sc.i;
        if (!read_payload()) return false;
        have_payload = true;
        sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != ']') return bad("malformed message: a sequence of more than two elements");
        ++sc.i;
    } else {
    if (sc.i >= sc.n || sc.p[sc.i] != '{') return bad("malformed message: expected an object");
    ++sc.i;
    sc.ws();
    if (sc.i < sc.n && sc.p[sc.i] == '}') ++sc.i;
    else while (true) {
        sc.ws();
        std::string key;
        if (!sc.string(&key)) return bad("malformed message: field name");
        sc.ws();
        if (sc.i >= sc.n || sc.p[sc.i] != ':') return bad("malformed message: expected ':'");
        ++sc.i;
        if (key == "result_id") {
            if (have_rid) return bad("malformed message: duplicate field `result_id`");
            have_rid = true;
            if (!read_rid()) return false;
        } else if (key == "payload") {
            if (have_payload) return bad("malformed message: duplicate field `payload`");
            have_payload = true;
            if (!read_payload()) return false;
        } else if (!sc.value(1)) {
            return bad("malformed message: value of an unknown field");
        }
        sc.ws();
        if (sc.i >= sc.n) return bad("malformed message: unexpected end");
        if (sc.p[sc.i] == ',') { ++sc.i; continue; }
        if (sc.p[sc.i] == '}') { ++sc.i; break; }
        return bad("malformed message: expected ',' or '}'");
    }
    }
    sc.ws();
    if (sc.i != sc.n) return bad("malformed message: trailing characters");
    if (!have_rid) return bad("malformed message: missing field `result_id`");
    if (!have_payload) return bad("malformed message: missing field `payload`");
    if (variant != "ScalarBatch" && variant != "PointBatch") return bad("unsupported payload variant");
    out.assign(8, 0);
    auto lit = [&](const char* t) { out.insert(out.end(), (const unsigned char*)t, (const unsigned char*)t + strlen(t)); };
    lit("{\"result_id\":");
    out.insert(out.end(), in.begin() + rid_a, in.begin() + rid_b);
    lit(",\"payload\":{\""); lit(variant.c_str()); lit("\":");
    if (!append_stripped(in.data(), arr_a, arr_b, out)) return bad("malformed message: whitespace inside a number");
    lit("}}");
    const u64 len = out.size() - 8;
    for (int k = 0; k < 8; ++k) out[k] = (unsigned char)(len >> (8 * k));
    return true;
}

const char* kind_name(int kind) { return kind == ARKMPC_WIRE_SCALAR_BATCH ? "ScalarBatch" : (kind == ARKMPC_WIRE_POINT_BATCH ? "PointBatch" : nullptr); }

WireHeader make_header(int kind, uint64_t result_id) {
    WireHeader h;
    int len = snprintf((char*)h.text, sizeof(h.text), "{\"result_id\":%llu,\"payload\":{\"%s\":[", (unsigned long long)result_id, kind_name(kind));
    h.len = (u32)len;
    return h;
}

#define DISPATCH_F(ctx, EXPR_F)                                                                 \
    switch ((ctx)->field_id) {                                                                  \
        case 0: { constexpr int F = 0; EXPR_F; } break;                                         \
        case 1: { constexpr int F = 1; EXPR_F; } break;                                         \
        case 2: { constexpr int F = 2; EXPR_F; } break;                                         \
        case 3: { constexpr int F = 3; EXPR_F; } break;                                         \
        case 4: { constexpr int F = 4; EXPR_F; } break;                                         \
        default: return ark_bad(ctx, "bad field id");                                           \
    }

// shared encoder: records are either given (recs32 != nullptr, caller layout mode) or produced from Montgomery scalars
int encode_locked(arkmpc_ctx* ctx, int kind, uint64_t result_id, size_t n, const unsigned char* recs32, const uint64_t* scalars, uint8_t* out_frame,
                  size_t out_cap, size_t* out_len);
int encode_impl(arkmpc_ctx* ctx, int kind, uint64_t result_id, size_t n, const unsigned char* recs32, const uint64_t* scalars, uint8_t* out_frame,
                size_t out_cap, size_t* out_len) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    return encode_locked(ctx, kind, result_id, n, recs32, scalars, out_frame, out_cap, out_len);
}
// (caller holds the context lock)
int encode_locked(arkmpc_ctx* ctx, int kind, uint64_t result_id, size_t n, const unsigned char* recs32, const uint64_t* scalars, uint8_t* out_frame,
                  size_t out_cap, size_t* out_len) {
    if (!kind_name(kind)) return ark_bad(ctx, "unknown payload kind");
    if (!out_len) return ark_bad(ctx, "null out_len");
    const WireHeader hdr = make_header(kind, result_id);
    const size_t bound = 8 + hdr.len + n * (size_t)WIRE_MAX_REC + 3;
    if (out_cap < bound) return ark_bad(ctx, "frame buffer smaller than arkmpc_wire_frame_bound(n)");
    Stage st(ctx);
    int ir = recs32 ? st.declare_in(recs32, n * 32) : -1, is = scalars ? st.declare_in(scalars, n * 32) : -1;
    int io = st.declare_out(out_frame, bound);
    size_t scan_bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const u64*)nullptr, (u64*)nullptr, n + 1, ctx->stream);
    if (e != hipSuccess) { ark_set_err(ctx, std::string("scan sizing: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
    int il = st.declare_scratch((n + 1) * 8), iof = st.declare_scratch((n + 1) * 8), it = st.declare_scratch(scan_bytes + 64),
        ic = st.declare_scratch(n * 32 + 32);
    if (st.commit()) return st.rc;
    hipStream_t s = ctx->stream;
    const unsigned char* recs = recs32 ? st.in<unsigned char>(ir) : st.scratch<unsigned char>(ic);
    if (!recs32 && n) DISPATCH_F(ctx, hipLaunchKernelGGL((k_wire_to_canonical<F>), dim3(blocks_for(n, WIRE_TPB)), dim3(WIRE_TPB), 0, s, n, st.in<u64>(is),
                                                         st.scratch<unsigned char>(ic)));
    hipLaunchKernelGGL(k_wire_lengths, dim3(blocks_for(n + 1, WIRE_TPB)), dim3(WIRE_TPB), 0, s, n, recs, st.scratch<u64>(il));
    e = hipcub::DeviceScan::ExclusiveSum(st.scratch<void>(it), scan_bytes, st.scratch<u64>(il), st.scratch<u64>(iof), n + 1, s);
    if (e != hipSuccess) { ark_set_err(ctx, std::string("scan: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
    hipLaunchKernelGGL(k_wire_render, dim3(n ? blocks_for(n, WIRE_TPB) : 1), dim3(WIRE_TPB), 0, s, n, recs, st.scratch<u64>(iof), hdr, st.out<unsigned char>(io));
    // the frame length is data dependent: read the scan total back
    u64* h_total = (u64*)ctx->h_flag;
    ARK_HIP(ctx, hipMemcpyAsync(h_total, st.scratch<u64>(iof) + n, 8, hipMemcpyDeviceToHost, s));
    ARK_HIP(ctx, hipStreamSynchronize(s));
    const size_t frame_len = 8 + hdr.len + (size_t)*h_total + 3;
    *out_len = frame_len;
    // host-buffer mode: only the bytes of the frame travel back
    if (ctx->host_buffers) st.oplan[io].bytes = frame_len;
    return st.finish();
}

}  // namespace

extern "C" {

int arkmpc_wire_frame_bound(size_t n, size_t* out_bytes) {
    if (!out_bytes) return ARKMPC_ERR_BAD_ARG;
    *out_bytes = 8 + 80 + n * (size_t)WIRE_MAX_REC + 3;
    return ARKMPC_OK;
}

int arkmpc_wire_encode_scalar_batch(arkmpc_ctx* ctx, uint64_t result_id, size_t n, const uint64_t* scalars, uint8_t* out_frame, size_t out_cap,
                                    size_t* out_len) {
    if (ctx && n && !scalars) return ark_bad(ctx, "null scalars");
    return encode_impl(ctx, ARKMPC_WIRE_SCALAR_BATCH, result_id, n, nullptr, n ? scalars : nullptr, out_frame, out_cap, out_len);
}

int arkmpc_wire_encode_bytes32(arkmpc_ctx* ctx, int kind, uint64_t result_id, size_t n, const uint8_t* records, uint8_t* out_frame, size_t out_cap,
                               size_t* out_len) {
    if (ctx && n && !records) return ark_bad(ctx, "null records");
    return encode_impl(ctx, kind, result_id, n, n ? records : nullptr, nullptr, out_frame, out_cap, out_len);
}

static int decode_strict(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, int want_kind, uint8_t* out_records, uint64_t* out_scalars,
                         size_t* out_n, uint64_t* out_result_id, int* out_kind);
// Entry: the strict GPU parser; a frame it rejects is re-read on the host (rare) with serde's object semantics and, if it is a message, parsed again.
static int decode_locked(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, int want_kind, uint8_t* out_records, uint64_t* out_scalars,
                         size_t* out_n, uint64_t* out_result_id, int* out_kind);
static int decode_impl(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, int want_kind, uint8_t* out_records, uint64_t* out_scalars,
                       size_t* out_n, uint64_t* out_result_id, int* out_kind) {
    if (!ctx) return ARKMPC_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    if (guard.rc) return guard.rc;
    return decode_locked(ctx, frame, frame_len, max_n, want_kind, out_records, out_scalars, out_n, out_result_id, out_kind);
}
// (caller holds the context lock)
static int decode_locked(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, int want_kind, uint8_t* out_records, uint64_t* out_scalars,
                         size_t* out_n, uint64_t* out_result_id, int* out_kind) {
    if (!frame || !out_n) return ark_bad(ctx, "null frame / out_n");
    if (frame_len < 8 + 30) return ark_bad(ctx, "frame too short");
    // The strict GPU parser runs FIRST: serde_json::to_vec -- what a reference peer sends -- emits the compact form, so the common frame
    // costs no extra scan, launch or synchronisation.
    if (!ctx->host_buffers && ((uintptr_t)frame & 15)) return ark_bad(ctx, "device pointer not 16-byte aligned");
    const int strict_rc = decode_strict(ctx, frame, frame_len, max_n, want_kind, out_records, out_scalars, out_n, out_result_id, out_kind);
    if (strict_rc != ARKMPC_ERR_BAD_ARG) return strict_rc;
    std::string strict_err;
    { std::lock_guard<std::mutex> lk(ctx->err_mu); strict_err = ctx->err; }
    // the frame is not in serde_json::to_vec's compact form.  It may still be a NetworkOutbound message as serde_json::from_slice reads one
    // (whitespace between tokens, fields in another order, unknown fields, escaped keys): re-read it on the host and, if so, canonicalise it.
    std::vector<unsigned char> raw(frame_len), norm;
    if (ctx->host_buffers) memcpy(raw.data(), frame, frame_len);
    else { ARK_HIP(ctx, hipMemcpyAsync(raw.data(), frame, frame_len, hipMemcpyDeviceToHost, ctx->stream)); ARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
    u64 declared = 0;
    for (int k = 0; k < 8; ++k) declared |= (u64)raw[k] << (8 * k);
    if (declared != frame_len - 8) return ark_bad(ctx, "length prefix does not match the frame");
    std::string why;
    if (!normalise_message(raw, norm, why)) { ark_set_err(ctx, why.empty() ? strict_err : why); return ARKMPC_ERR_BAD_ARG; }
    if (norm.size() == raw.size() && memcmp(norm.data(), raw.data(), raw.size()) == 0) { ark_set_err(ctx, strict_err); return strict_rc; }   // already canonical: the strict verdict stands
    if (norm.size() < 8 + 30) return ark_bad(ctx, "frame too short");
    if (ctx->host_buffers) return decode_strict(ctx, norm.data(), norm.size(), max_n, want_kind, out_records, out_scalars, out_n, out_result_id, out_kind);
    void* dcopy = nullptr;
    ARK_HIP(ctx, hipMalloc(&dcopy, norm.size() + 16));
    int rc = ARKMPC_OK;
    if (hipMemcpyAsync(dcopy, norm.data(), norm.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ARKMPC_ERR_HIP;
    if (!rc) rc = decode_strict(ctx, (const uint8_t*)dcopy, norm.size(), max_n, want_kind, out_records, out_scalars, out_n, out_result_id, out_kind);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dcopy);
    return rc;
}
// Parses the frame header on the host (it is < 100 bytes), the body on the device.  Compact form only (no whitespace).
static int decode_strict(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, int want_kind, uint8_t* out_records, uint64_t* out_scalars,
                         size_t* out_n, uint64_t* out_result_id, int* out_kind) {
    // header bytes to the host
    unsigned char head[112];
    const size_t hl = frame_len < sizeof(head) ? frame_len : sizeof(head);
    unsigned char tail[3];
    if (ctx->host_buffers) {
        memcpy(head, frame, hl);
        memcpy(tail, frame + frame_len - 3, 3);
    } else {
        ARK_HIP(ctx, hipMemcpyAsync(ctx->h_flag, frame, hl < 48 ? hl : 48, hipMemcpyDeviceToHost, ctx->stream));   // h_flag is 64 bytes
        ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(head, ctx->h_flag, hl < 48 ? hl : 48);
        if (hl > 48) {
            ARK_HIP(ctx, hipMemcpyAsync(ctx->h_flag, frame + 48, hl - 48, hipMemcpyDeviceToHost, ctx->stream));
            ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(head + 48, ctx->h_flag, hl - 48);
        }
        ARK_HIP(ctx, hipMemcpyAsync(ctx->h_flag, frame + frame_len - 3, 3, hipMemcpyDeviceToHost, ctx->stream));
        ARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(tail, ctx->h_flag, 3);
    }
    u64 declared = 0;
    for (int k = 0; k < 8; ++k) declared |= (u64)head[k] << (8 * k);
    if (declared != frame_len - 8) return ark_bad(ctx, "length prefix does not match the frame");
    size_t p = 8;
    auto expect = [&](const char* lit) -> bool {
        const size_t l = strlen(lit);
        if (p + l > hl || memcmp(head + p, lit, l) != 0) return false;
        p += l;
        return true;
    };
    if (!expect("{\"result_id\":")) return ark_bad(ctx, "malformed message: result_id");
    u64 rid = 0;
    size_t nd = 0;
    bool rid_overflow = false;
    while (p < hl && head[p] >= '0' && head[p] <= '9' && nd < 21) {
        const u64 dgt = (u64)(head[p] - '0');
        if (rid > (~(u64)0 - dgt) / 10) rid_overflow = true;          // usize on the reference side: must fit 64 bits
        rid = rid * 10 + dgt; ++p; ++nd;
    }
    if (nd == 0 || nd > 20 || rid_overflow || (nd > 1 && head[p - nd] == '0')) return ark_bad(ctx, "malformed message: result_id");
    if (!expect(",\"payload\":{\"")) return ark_bad(ctx, "malformed message: payload");
    int kind = -1;
    if (expect("ScalarBatch\":[")) kind = ARKMPC_WIRE_SCALAR_BATCH;
    else if (expect("PointBatch\":[")) kind = ARKMPC_WIRE_POINT_BATCH;
    else return ark_bad(ctx, "unsupported payload variant");
    if (want_kind >= 0 && kind != want_kind) return ark_bad(ctx, "payload variant differs from the expected one");
    if (memcmp(tail, "]}}", 3) != 0 || frame_len < p + 3) return ark_bad(ctx, "malformed message: trailer");
    const size_t body_off = p, body_len = frame_len - 3 - p;
    if (out_result_id) *out_result_id = rid;
    if (out_kind) *out_kind = kind;

    Stage st(ctx);
    int ifr = st.declare_in(frame, frame_len);
    int ior = out_records ? st.declare_out(out_records, max_n * 32) : -1, ios = out_scalars ? st.declare_out(out_scalars, max_n * 32) : -1;
    const size_t v0 = body_off / 16, v1 = (body_off + body_len + 15) / 16;      // aligned 16-byte vectors covering the body
    const u32 nblocks = (u32)((v1 - v0 + WIRE_TPB - 1) / WIRE_TPB);
    size_t scan_bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const u32*)nullptr, (u32*)nullptr, nblocks + 1, ctx->stream);
    if (e != hipSuccess) { ark_set_err(ctx, std::string("scan sizing: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
    int ic = st.declare_scratch(((size_t)nblocks + 1) * 4), io = st.declare_scratch(((size_t)nblocks + 1) * 4), it = st.declare_scratch(scan_bytes + 64),
        ip = st.declare_scratch((max_n + 1) * 8), irc = st.declare_scratch(max_n * 32 + 32);
    if (st.commit()) return st.rc;
    hipStream_t s = ctx->stream;
    const unsigned char* fr = st.in<unsigned char>(ifr);
    const size_t lo = body_off, hi = body_off + body_len;
    size_t n = 0;
    if (body_len) {
        hipLaunchKernelGGL(k_wire_count, dim3(nblocks), dim3(WIRE_TPB), 0, s, fr, v0, lo, hi, st.scratch<u32>(ic), nblocks);
        e = hipcub::DeviceScan::ExclusiveSum(st.scratch<void>(it), scan_bytes, st.scratch<u32>(ic), st.scratch<u32>(io), nblocks + 1, s);
        if (e != hipSuccess) { ark_set_err(ctx, std::string("scan: ") + hipGetErrorString(e)); return ARKMPC_ERR_HIP; }
        ARK_HIP(ctx, hipMemcpyAsync(ctx->h_flag, st.scratch<u32>(io) + nblocks, 4, hipMemcpyDeviceToHost, s));
        ARK_HIP(ctx, hipStreamSynchronize(s));
        n = (size_t)*(u32*)ctx->h_flag;
        if (n == 0) return ark_bad(ctx, "malformed message: body");
        if (n > max_n) { *out_n = n; return ark_bad(ctx, "more elements than the output buffer holds"); }
        ARK_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_wire_positions, dim3(nblocks), dim3(WIRE_TPB), 0, s, fr, v0, lo, hi, st.scratch<u32>(io), st.scratch<u64>(ip), max_n);
        unsigned char* recs = out_records ? st.out<unsigned char>(ior) : st.scratch<unsigned char>(irc);
        hipLaunchKernelGGL(k_wire_parse, dim3(blocks_for(n, WIRE_TPB)), dim3(WIRE_TPB), 0, s, n, lo, hi, fr, st.scratch<u64>(ip), recs, ctx->d_flag);
        if (out_scalars) DISPATCH_F(ctx, hipLaunchKernelGGL((k_wire_scalars<F>), dim3(blocks_for(n, WIRE_TPB)), dim3(WIRE_TPB), 0, s, n, recs,
                                                            st.out<u64>(ios), ctx->d_flag));
        ARK_HIP(ctx, hipMemcpyAsync(ctx->h_flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        ARK_HIP(ctx, hipStreamSynchronize(s));
        const int err = *ctx->h_flag;
        if (err & WIRE_E_SYNTAX) return ark_bad(ctx, "malformed message: element syntax");
        if (err & WIRE_E_RANGE) return ark_bad(ctx, "element out of range (byte > 255 or scalar >= modulus)");
    }
    *out_n = n;
    if (ctx->host_buffers) {
        if (ior >= 0) st.oplan[ior].bytes = n * 32;
        if (ios >= 0) st.oplan[ios].bytes = n * 32;
    }
    return st.finish();
}

int arkmpc_wire_decode_scalar_batch(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, uint64_t* out_scalars, size_t* out_n,
                                    uint64_t* out_result_id) {
    if (ctx && max_n && !out_scalars) return ark_bad(ctx, "null output");
    return decode_impl(ctx, frame, frame_len, max_n, ARKMPC_WIRE_SCALAR_BATCH, nullptr, out_scalars, out_n, out_result_id, nullptr);
}
int arkmpc_wire_decode_bytes32(arkmpc_ctx* ctx, const uint8_t* frame, size_t frame_len, size_t max_n, uint8_t* out_records, size_t* out_n,
                               uint64_t* out_result_id, int* out_kind) {
    if (ctx && max_n && !out_records) return ark_bad(ctx, "null output");
    return decode_impl(ctx, frame, frame_len, max_n, -1, out_records, nullptr, out_n, out_result_id, out_kind);
}

}  // extern "C"

// The codec on DEVICE buffers whatever the context's buffer mode, for callers inside the library that hold the context lock (the wire form of the
// streaming sessions, csrc/arkmpc_stream.inc): scalars / frame are device pointers, *out_len / *out_n come back after a synchronisation.
__attribute__((visibility("hidden"))) int ark_wire_encode_scalars_device(arkmpc_ctx* ctx, uint64_t result_id, size_t n, const uint64_t* d_scalars, uint8_t* d_frame,
                                                                         size_t cap, size_t* out_len) {
    const bool mode = ctx->host_buffers;
    ctx->host_buffers = false;
    const int rc = encode_locked(ctx, ARKMPC_WIRE_SCALAR_BATCH, result_id, n, nullptr, n ? d_scalars : nullptr, d_frame, cap, out_len);
    ctx->host_buffers = mode;
    return rc;
}
__attribute__((visibility("hidden"))) int ark_wire_decode_scalars_device(arkmpc_ctx* ctx, const uint8_t* d_frame, size_t frame_len, size_t max_n, uint64_t* d_scalars,
                                                                         size_t* out_n, uint64_t* out_result_id) {
    const bool mode = ctx->host_buffers;
    ctx->host_buffers = false;
    const int rc = decode_locked(ctx, d_frame, frame_len, max_n, ARKMPC_WIRE_SCALAR_BATCH, nullptr, d_scalars, out_n, out_result_id, nullptr);
    ctx->host_buffers = mode;
    return rc;
}
