// capi_serialize.hip -- part of libakp.so (implementation of include/akp.h): the ark-serialize byte formats of the structs that
// cross the boundary, for hosts that are not Python.  Host only: no device, no context needed.
// Product code.  Never includes, links or calls anything under oracle/.
//
// What the reference derives (`#[derive(CanonicalSerialize, CanonicalDeserialize)]`, fields in declaration order):
//   PoseidonConfig            sponge/poseidon/mod.rs:26-45
//   pedersen::Parameters      crh/pedersen/mod.rs:28-31        bowe_hopwood::Parameters   crh/bowe_hopwood/mod.rs:33-37
//   Path                      merkle_tree/mod.rs:139-152       MultiPath                  merkle_tree/mod.rs:239-254
// Leaf encodings (ark-serialize / ark-ff / ark-ec, un-vendored dependencies of the reference): usize and u64 as 8 bytes LE,
// Vec<T> as a u64 length + elements, Fp as 32 bytes LE of the canonical integer, a twisted-Edwards affine point as x || y
// (uncompressed) or as y with the sign of x in the top bit of the last byte (compressed; "negative" = x > (p - 1) / 2), a
// projective point as its affine form.  Reading validates like `Validate::Yes` unless told otherwise: canonical field
// elements always; points on the curve and in the prime-order subgroup when `validate` is set.
#include "capi_internal.hpp"

namespace {
// ---- Fq helpers on the host (Montgomery form, fr.hpp) ---------------------------------------------------------------------
const uint64_t P_WORDS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
Fr fr_small(uint64_t v) {
    Fr c = fr_zero();
    c.l[0] = (u32)v;
    c.l[1] = (u32)(v >> 32);
    return fr_to_mont(c);
}
// a^e, e = 256-bit little-endian words
Fr fr_pow_words(const Fr& a, const uint64_t e[4]) {
    Fr r = fr_one();
    for (int i = 255; i >= 0; --i) {
        r = fr_sqr(r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fr_mul(r, a);
    }
    return r;
}
void words_shr(const uint64_t in[4], unsigned k, uint64_t out[4]) {  // k < 64
    for (int i = 0; i < 4; ++i) out[i] = (in[i] >> k) | (k && i < 3 ? in[i + 1] << (64 - k) : 0);
}
struct FqConsts {
    Fr d;              // Jubjub d = -(10240 / 10241)
    Fr ts_c;           // z^t for a non-residue z, t = (p - 1) / 2^32
    uint64_t half[4];  // (p - 1) / 2: Euler exponent and the sign threshold of TEFlags
    uint64_t t[4];     // (p - 1) / 2^32
    uint64_t t1h[4];   // (t + 1) / 2
    FqConsts() {
        d = fr_neg(fr_mul(fr_small(10240), fr_inv(fr_small(10241))));
        uint64_t pm1[4] = {P_WORDS[0] - 1, P_WORDS[1], P_WORDS[2], P_WORDS[3]};
        words_shr(pm1, 1, half);
        uint64_t tmp[4];
        words_shr(pm1, 32, t);  // p - 1 = 2^32 * t, t odd
        tmp[0] = t[0] + 1;      // t is odd: no carry out of the low word unless it is all ones (it is not)
        tmp[1] = t[1]; tmp[2] = t[2]; tmp[3] = t[3];
        words_shr(tmp, 1, t1h);
        Fr z = fr_small(2);
        for (uint64_t k = 2;; ++k) {  // smallest non-residue
            z = fr_small(k);
            if (!fr_eq(fr_pow_words(z, half), fr_one())) break;
        }
        ts_c = fr_pow_words(z, t);
    }
};
const FqConsts& fq() {
    static const FqConsts c;
    return c;
}
// square root by Tonelli-Shanks (p - 1 = 2^32 t); false for a non-residue
bool fr_sqrt(const Fr& a, Fr& out) {
    if (fr_is_zero(a)) { out = a; return true; }
    const FqConsts& k = fq();
    if (!fr_eq(fr_pow_words(a, k.half), fr_one())) return false;
    u32 m = 32;
    Fr c = k.ts_c, tt = fr_pow_words(a, k.t), r = fr_pow_words(a, k.t1h);
    const Fr one = fr_one();
    while (!fr_eq(tt, one)) {
        u32 i = 0;
        Fr u = tt;
        while (!fr_eq(u, one)) { u = fr_sqr(u); ++i; }
        Fr b = c;
        for (u32 j = 0; j + i + 1 < m; ++j) b = fr_sqr(b);
        m = i;
        c = fr_sqr(b);
        tt = fr_mul(tt, c);
        r = fr_mul(r, b);
    }
    out = r;
    return true;
}
bool words_gt(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return false;
}
// wire (Montgomery) -> 32 canonical little-endian bytes
void fq_write(const uint64_t* wire, uint8_t* out) {
    uint64_t c[4];
    fr_to_words(fr_from_mont(fr_from_words(wire)), c);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(c[i] >> (8 * b));
}
bool fq_x_is_negative(const uint64_t* wire) {  // TEFlags::from_x_coordinate: x > -x  <=>  x > (p - 1) / 2
    uint64_t c[4];
    fr_to_words(fr_from_mont(fr_from_words(wire)), c);
    return words_gt(c, fq().half);
}
void words_of_bytes(const uint8_t* in, uint64_t c[4]) {
    for (int i = 0; i < 4; ++i) {
        c[i] = 0;
        for (int b = 0; b < 8; ++b) c[i] |= (uint64_t)in[8 * i + b] << (8 * b);
    }
}
// 32 canonical bytes -> wire; false when the integer is >= p
bool fq_read(const uint8_t* in, uint64_t* wire) {
    uint64_t c[4];
    words_of_bytes(in, c);
    if (!fr_words_reduced(c)) return false;
    fr_to_words(fr_to_mont(fr_from_words(c)), wire);
    return true;
}
// ---- Jubjub on the host: the two checks of Validate::Yes -------------------------------------------------------------------
struct PtE { Fr X, Y, Z, T; };
PtE pt_add(const PtE& p, const PtE& q) {  // unified add-2008-hwcd, a = -1 (complete: d is a non-square)
    const Fr a = fr_mul(p.X, q.X), b = fr_mul(p.Y, q.Y), c = fr_mul(fr_mul(fq().d, p.T), q.T), dd = fr_mul(p.Z, q.Z);
    const Fr e = fr_sub(fr_sub(fr_mul(fr_add(p.X, p.Y), fr_add(q.X, q.Y)), a), b), f = fr_sub(dd, c), g = fr_add(dd, c), h = fr_add(b, a);
    return PtE{fr_mul(e, f), fr_mul(g, h), fr_mul(f, g), fr_mul(e, h)};
}
bool pt_on_curve(const Fr& x, const Fr& y) {
    const Fr x2 = fr_sqr(x), y2 = fr_sqr(y);
    return fr_eq(fr_sub(y2, x2), fr_add(fr_one(), fr_mul(fq().d, fr_mul(x2, y2))));
}
bool pt_in_subgroup(const Fr& x, const Fr& y) {  // r * P == O
    static const uint64_t R[4] = {0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL};
    PtE acc{fr_zero(), fr_one(), fr_one(), fr_zero()};
    const PtE base{x, y, fr_one(), fr_mul(x, y)};
    for (int i = 251; i >= 0; --i) {
        acc = pt_add(acc, acc);
        if ((R[i >> 6] >> (i & 63)) & 1) acc = pt_add(acc, base);
    }
    return fr_is_zero(acc.X) && fr_eq(acc.Y, acc.Z);
}
int32_t point_check(const uint64_t* xy_wire, const char* what, size_t index) {
    const Fr x = fr_from_words(xy_wire), y = fr_from_words(xy_wire + 4);
    if (!pt_on_curve(x, y)) return fail(AKP_ERR_BAD_PARAMS, "%s: point %zu is not on the curve", what, index);
    if (!pt_in_subgroup(x, y)) return fail(AKP_ERR_BAD_PARAMS, "%s: point %zu is not in the prime-order subgroup", what, index);
    return AKP_OK;
}

// ---- cursor-style writer / reader -------------------------------------------------------------------------------------------
struct Writer {  // counts always, writes while the capacity lasts: one pass serves the size query and the real call
    uint8_t* out;
    size_t cap, len = 0;
    bool overflow = false;
    Writer(uint8_t* o, size_t c) : out(o), cap(o ? c : 0) { if (!o) overflow = true; }
    uint8_t* room(size_t n) {
        uint8_t* p = (!overflow && len + n <= cap) ? out + len : nullptr;
        if (!p) overflow = true;
        len += n;
        return p;
    }
    void u64(uint64_t v) {
        if (uint8_t* p = room(8))
            for (int b = 0; b < 8; ++b) p[b] = (uint8_t)(v >> (8 * b));
    }
    void fq(const uint64_t* wire) {
        if (uint8_t* p = room(32)) fq_write(wire, p);
    }
    void digest(const uint64_t* wire, u32 fe, bool compress) {
        if (fe == 1) return fq(wire);
        if (!compress) { fq(wire); fq(wire + 4); return; }
        if (uint8_t* p = room(32)) {
            fq_write(wire + 4, p);
            if (fq_x_is_negative(wire)) p[31] |= 0x80;
        }
    }
};
int32_t finish(const Writer& w, uint8_t* out, size_t* out_len, const char* what) {
    if (out_len) *out_len = w.len;
    if (out && w.overflow) return fail(AKP_ERR_BAD_LENGTH, "%s: %zu bytes needed, the buffer holds %zu", what, w.len, w.cap);
    return AKP_OK;
}
struct Reader {
    const uint8_t* in;
    size_t len, at = 0;
    const char* what;
    int32_t rc = AKP_OK;
    Reader(const uint8_t* i, size_t l, const char* w) : in(i), len(l), what(w) {}
    const uint8_t* take(size_t n) {
        if (rc) return nullptr;
        if (n > len - at) { rc = fail(AKP_ERR_BAD_LENGTH, "%s: unexpected end of input at byte %zu", what, at); return nullptr; }
        const uint8_t* p = in + at;
        at += n;
        return p;
    }
    uint64_t u64() {
        const uint8_t* p = take(8);
        uint64_t v = 0;
        if (p) for (int b = 0; b < 8; ++b) v |= (uint64_t)p[b] << (8 * b);
        return v;
    }
    // a Vec length, checked against the bytes that are left so that a corrupt prefix cannot ask for gigabytes
    size_t count(size_t min_item_bytes) {
        const uint64_t n = u64();
        if (!rc && min_item_bytes && n > (len - at) / min_item_bytes) rc = fail(AKP_ERR_BAD_LENGTH, "%s: length prefix %llu exceeds the input", what,
                (unsigned long long)n);
        return rc ? 0 : (size_t)n;
    }
    void fq(uint64_t* wire) {  // wire may be NULL (size query): the bytes are still checked
        const uint8_t* p = take(32);
        uint64_t tmp[4];
        if (p && !fq_read(p, wire ? wire : tmp)) rc = fail(AKP_ERR_BAD_PARAMS, "%s: field element at byte %zu is not canonical", what, at - 32);
    }
    void digest(uint64_t* wire, u32 fe, bool compress, bool validate, size_t index) {
        uint64_t tmp[8];
        uint64_t* w = wire ? wire : tmp;
        if (fe == 1) return fq(w);
        if (!compress) {
            fq(w);
            fq(w + 4);
        } else {
            const uint8_t* p = take(32);
            if (!p) return;
            uint8_t raw[32];
            memcpy(raw, p, 32);
            const bool negative = (raw[31] & 0x80) != 0;
            raw[31] &= 0x7f;
            if (!fq_read(raw, w + 4)) { rc = fail(AKP_ERR_BAD_PARAMS, "%s: y coordinate at byte %zu is not canonical", what, at - 32); return; }
            // x^2 = (y^2 - 1) / (1 + d y^2)
            const Fr y = fr_from_words(w + 4), y2 = fr_sqr(y);
            Fr x;
            if (!fr_sqrt(fr_mul(fr_sub(y2, fr_one()), fr_inv(fr_add(fr_one(), fr_mul(fq_consts_d(), y2)))), x)) {
                rc = fail(AKP_ERR_BAD_PARAMS, "%s: no point of the curve has the y coordinate at byte %zu", what, at - 32);
                return;
            }
            fr_to_words(x, w);
            if (fq_x_is_negative(w) != negative) fr_to_words(fr_neg(x), w);
        }
        if (!rc && validate) rc = point_check(w, what, index);
    }
    static Fr fq_consts_d() { return ::fq().d; }
    int32_t done() {
        if (!rc && at != len) rc = fail(AKP_ERR_BAD_LENGTH, "%s: %zu trailing bytes", what, len - at);
        return rc;
    }
};
inline size_t digest_bytes(u32 fe, bool compress) { return (fe == 1 || compress) ? 32 : 64; }
int32_t check_fe(u32 fe, const char* what) { return (fe == 1 || fe == 2) ? AKP_OK : fail(AKP_ERR_BAD_PARAMS, "%s: fe_per_digest must be 1 or 2", what); }
}  // namespace

// ---- digests (LeafDigest / InnerDigest / CRH outputs): n of them back to back, no length prefix ---------------------------------
extern "C" int32_t akp_serialize_digests(const uint64_t* digests, size_t n, uint32_t fe, int32_t compress, uint8_t* out, size_t out_cap,
        size_t* out_len) {
    if (int32_t rc = check_fe(fe, "akp_serialize_digests")) return rc;
    if (n && !digests) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_digests: digests is NULL");
    Writer w(out, out_cap);
    for (size_t i = 0; i < n; ++i) w.digest(digests + i * 4 * fe, fe, compress != 0);
    return finish(w, out, out_len, "akp_serialize_digests");
}
extern "C" int32_t akp_deserialize_digests(const uint8_t* in, size_t in_len, size_t n, uint32_t fe, int32_t compress, int32_t validate,
        uint64_t* digests) {
    if (int32_t rc = check_fe(fe, "akp_deserialize_digests")) return rc;
    if (n && (!in || !digests)) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_digests: NULL buffer");
    Reader r(in, in_len, "akp_deserialize_digests");
    for (size_t i = 0; i < n && !r.rc; ++i) r.digest(digests + i * 4 * fe, fe, compress != 0, validate != 0, i);
    return r.done();
}

// ---- PoseidonConfig (sponge/poseidon/mod.rs:26-45): both modes write the same bytes ---------------------------------------------
extern "C" int32_t akp_serialize_poseidon_config(const akp_poseidon* p, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_poseidon_config: params is NULL");
    const PoseidonDims& d = p->dims;
    const size_t rounds = (size_t)d.full_rounds + d.partial_rounds;
    Writer w(out, out_cap);
    w.u64(d.full_rounds);
    w.u64(d.partial_rounds);
    w.u64(d.alpha);
    uint64_t tmp[4];
    w.u64(rounds);
    for (size_t r = 0; r < rounds; ++r) {
        w.u64(d.t);
        for (u32 i = 0; i < d.t; ++i) { fr_to_words(p->ark[r * d.t + i], tmp); w.fq(tmp); }
    }
    w.u64(d.t);
    for (u32 r = 0; r < d.t; ++r) {
        w.u64(d.t);
        for (u32 i = 0; i < d.t; ++i) { fr_to_words(p->mds[(size_t)r * d.t + i], tmp); w.fq(tmp); }
    }
    w.u64(d.rate);
    w.u64(d.capacity);
    return finish(w, out, out_len, "akp_serialize_poseidon_config");
}
extern "C" int32_t akp_deserialize_poseidon_config(akp_ctx* ctx, const uint8_t* in, size_t in_len, akp_poseidon** out) {
    if (!in || !out) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_poseidon_config: NULL argument");
    Reader r(in, in_len, "akp_deserialize_poseidon_config");
    const uint64_t full = r.u64(), partial = r.u64(), alpha = r.u64();
    std::vector<uint64_t> ark, mds;
    size_t ark_cols = 0, mds_cols = 0;
    bool ragged = false;
    auto matrix = [&](std::vector<uint64_t>& m, size_t& cols) -> size_t {
        const size_t rows = r.count(8);
        for (size_t i = 0; i < rows && !r.rc; ++i) {
            const size_t c = r.count(32);
            if (i == 0) cols = c;
            else if (c != cols) ragged = true;
            const size_t base = m.size();
            m.resize(base + 4 * c);
            for (size_t k = 0; k < c && !r.rc; ++k) r.fq(m.data() + base + 4 * k);
        }
        return rows;
    };
    const size_t ark_rows = matrix(ark, ark_cols), mds_rows = matrix(mds, mds_cols);
    const uint64_t rate = r.u64(), capacity = r.u64();
    if (int32_t rc = r.done()) return rc;
    // the bytes are a valid PoseidonConfig for ark-serialize whatever the shapes; a handle needs the shapes PoseidonConfig::new
    // asserts (:191-217)
    const uint64_t t = rate + capacity;
    if (ragged || rate > AKP_MAX_T || capacity > AKP_MAX_T || ark_cols != t || mds_cols != t || mds_rows != t || ark_rows != full + partial
        || full > 0xffffffffu || partial > 0xffffffffu)
        return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_poseidon_config: ark is %zu x %zu, mds %zu x %zu for t = %llu, %llu + %llu rounds", ark_rows,
                ark_cols, mds_rows, mds_cols, (unsigned long long)t, (unsigned long long)full, (unsigned long long)partial);
    return akp_poseidon_params_create(ctx, (uint32_t)full, (uint32_t)partial, alpha, (uint32_t)rate, (uint32_t)capacity, ark.data(), mds.data(), out);
}

// ---- Parameters { generators: Vec<Vec<C>> } (crh/pedersen/mod.rs:28-31, crh/bowe_hopwood/mod.rs:33-37) -----------------------
extern "C" int32_t akp_serialize_te_parameters(const uint64_t* gens, uint32_t window_size, uint32_t num_windows, int32_t compress, uint8_t* out,
        size_t out_cap, size_t* out_len) {
    if (!gens && (size_t)window_size * num_windows) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_te_parameters: generators is NULL");
    Writer w(out, out_cap);
    w.u64(num_windows);
    for (size_t i = 0; i < num_windows; ++i) {
        w.u64(window_size);
        for (size_t j = 0; j < window_size; ++j) w.digest(gens + (i * window_size + j) * 8, 2, compress != 0);
    }
    return finish(w, out, out_len, "akp_serialize_te_parameters");
}
extern "C" int32_t akp_deserialize_te_parameters(const uint8_t* in, size_t in_len, int32_t compress, int32_t validate, uint64_t* gens,
        size_t cap_points, uint32_t* window_size, uint32_t* num_windows) {
    if (!in) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_te_parameters: in is NULL");
    Reader r(in, in_len, "akp_deserialize_te_parameters");
    const size_t per = digest_bytes(2, compress != 0);
    const size_t rows = r.count(8);
    size_t cols = 0, written = 0;
    for (size_t i = 0; i < rows && !r.rc; ++i) {
        const size_t c = r.count(per);
        if (i == 0) cols = c;
        else if (c != cols) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_te_parameters: window %zu holds %zu generators, window 0 holds %zu", i, c, cols);
        for (size_t j = 0; j < c && !r.rc; ++j, ++written) {
            if (gens && written >= cap_points) return fail(AKP_ERR_BAD_LENGTH, "akp_deserialize_te_parameters: more than %zu generators", cap_points);
            r.digest(gens ? gens + written * 8 : nullptr, 2, compress != 0, gens != nullptr && validate != 0, written);
        }
    }
    if (int32_t rc = r.done()) return rc;
    if (rows > 0xffffffffu || cols > 0xffffffffu) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_te_parameters: window too large");
    if (num_windows) *num_windows = (uint32_t)rows;
    if (window_size) *window_size = (uint32_t)cols;
    return AKP_OK;
}

// Path / MultiPath carry TWO digest types: LeafDigest (leaf_sibling_hash) and InnerDigest (the authentication path); a Config may give them
// different widths (a Pedersen leaf hash whose affine point feeds, through a DigestConverter, a field-element two-to-one hash).  The
// fe_per_digest argument of the four entry points below is AKP_FE_PAIR(leaf_fe, inner_fe) = leaf_fe << 8 | inner_fe; a plain 1 or 2
// means both (every configuration of the reference's tests).
static int32_t split_fe(uint32_t* fe_inner, u32* fe_leaf, const char* what) {
    const u32 v = *fe_inner, inner = v & 0xffu, leaf = (v >> 8) ? (v >> 8) : inner;
    if (v >> 16) return fail(AKP_ERR_BAD_PARAMS, "%s: fe_per_digest %u is not 1, 2 or AKP_FE_PAIR(leaf, inner)", what, v);
    if (int32_t rc = check_fe(inner, what)) return rc;
    if (int32_t rc = check_fe(leaf, what)) return rc;
    *fe_inner = inner;
    *fe_leaf = leaf;
    return AKP_OK;
}

// ---- Path (merkle_tree/mod.rs:139-152) --------------------------------------------------------------------------------------------
extern "C" int32_t akp_serialize_path(const uint64_t* leaf_sibling_hash, const uint64_t* auth_path, size_t depth, uint64_t leaf_index, uint32_t fe,
        int32_t compress, uint8_t* out, size_t out_cap, size_t* out_len) {
    u32 lfe = 0;
    if (int32_t rc = split_fe(&fe, &lfe, "akp_serialize_path")) return rc;
    if (!leaf_sibling_hash || (depth && !auth_path)) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_path: NULL buffer");
    Writer w(out, out_cap);
    w.digest(leaf_sibling_hash, lfe, compress != 0);
    w.u64(depth);
    for (size_t j = 0; j < depth; ++j) w.digest(auth_path + j * 4 * fe, fe, compress != 0);
    w.u64(leaf_index);
    return finish(w, out, out_len, "akp_serialize_path");
}
extern "C" int32_t akp_deserialize_path(const uint8_t* in, size_t in_len, uint32_t fe, int32_t compress, int32_t validate, uint64_t* leaf_sibling_hash,
        uint64_t* auth_path, size_t auth_cap, size_t* depth, uint64_t* leaf_index) {
    u32 lfe = 0;
    if (int32_t rc = split_fe(&fe, &lfe, "akp_deserialize_path")) return rc;
    if (!in) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_path: in is NULL");
    Reader r(in, in_len, "akp_deserialize_path");
    const bool fill = leaf_sibling_hash != nullptr;
    r.digest(leaf_sibling_hash, lfe, compress != 0, fill && validate != 0, 0);
    const size_t d = r.count(digest_bytes(fe, compress != 0));
    if (!r.rc && fill && d > auth_cap) return fail(AKP_ERR_BAD_LENGTH, "akp_deserialize_path: auth_path holds %zu digests, the buffer %zu", d, auth_cap);
    if (!r.rc && fill && d && !auth_path) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_path: auth_path is NULL");
    for (size_t j = 0; j < d && !r.rc; ++j) r.digest(fill ? auth_path + j * 4 * fe : nullptr, fe, compress != 0, fill && validate != 0, j + 1);
    const uint64_t idx = r.u64();
    if (int32_t rc = r.done()) return rc;
    if (depth) *depth = d;
    if (leaf_index) *leaf_index = idx;
    return AKP_OK;
}

// ---- MultiPath (merkle_tree/mod.rs:239-254) ----------------------------------------------------------------------------------------
// Flat form: m paths; suffixes concatenated (what akp_merkle_multipath_encode writes); suffix i holds suffix_lengths[i] digests, or
// depth - prefix_lengths[i] when suffix_lengths is NULL.
extern "C" int32_t akp_serialize_multipath(const uint64_t* leaf_siblings_hashes, const uint64_t* prefix_lengths, const uint64_t* suffix_lengths,
        const uint64_t* suffixes, const uint64_t* leaf_indexes, size_t m, size_t depth, uint32_t fe, int32_t compress, uint8_t* out, size_t out_cap,
        size_t* out_len) {
    u32 lfe = 0;
    if (int32_t rc = split_fe(&fe, &lfe, "akp_serialize_multipath")) return rc;
    if (m && (!leaf_siblings_hashes || !prefix_lengths || !leaf_indexes)) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_multipath: NULL buffer");
    Writer w(out, out_cap);
    w.u64(m);
    for (size_t i = 0; i < m; ++i) w.digest(leaf_siblings_hashes + i * 4 * lfe, lfe, compress != 0);
    w.u64(m);
    for (size_t i = 0; i < m; ++i) w.u64(prefix_lengths[i]);
    w.u64(m);
    size_t at = 0;
    for (size_t i = 0; i < m; ++i) {
        if (!suffix_lengths && prefix_lengths[i] > depth) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_multipath: path %zu: prefix length %llu > depth %zu", i,
                (unsigned long long)prefix_lengths[i], depth);
        const size_t k = suffix_lengths ? (size_t)suffix_lengths[i] : depth - (size_t)prefix_lengths[i];
        if (k && !suffixes) return fail(AKP_ERR_BAD_PARAMS, "akp_serialize_multipath: suffixes is NULL");
        w.u64(k);
        for (size_t j = 0; j < k; ++j, ++at) w.digest(suffixes + at * 4 * fe, fe, compress != 0);
    }
    w.u64(m);
    for (size_t i = 0; i < m; ++i) w.u64(leaf_indexes[i]);
    return finish(w, out, out_len, "akp_serialize_multipath");
}
// Two passes: with leaf_siblings_hashes == NULL only *m and *n_suffix_digests are produced (and the bytes checked, without the
// point validation); with buffers of m_cap paths / suffix_cap digests everything is filled.  The reference's MultiPath allows its
// four vectors to have different lengths; the flat form does not, and such input is AKP_ERR_BAD_PARAMS.
extern "C" int32_t akp_deserialize_multipath(const uint8_t* in, size_t in_len, uint32_t fe, int32_t compress, int32_t validate, size_t* m_out,
        size_t* n_suffix_out, uint64_t* leaf_siblings_hashes, uint64_t* prefix_lengths, uint64_t* suffix_lengths, uint64_t* suffixes,
        uint64_t* leaf_indexes, size_t m_cap, size_t suffix_cap) {
    u32 lfe = 0;
    if (int32_t rc = split_fe(&fe, &lfe, "akp_deserialize_multipath")) return rc;
    if (!in) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: in is NULL");
    const bool fill = leaf_siblings_hashes != nullptr;
    if (fill && (!prefix_lengths || !suffix_lengths || !leaf_indexes)) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: NULL buffer");
    Reader r(in, in_len, "akp_deserialize_multipath");
    const size_t per = digest_bytes(fe, compress != 0);
    const bool val = fill && validate != 0;
    const size_t m = r.count(digest_bytes(lfe, compress != 0));
    if (!r.rc && fill && m > m_cap) return fail(AKP_ERR_BAD_LENGTH, "akp_deserialize_multipath: %zu paths, the buffers hold %zu", m, m_cap);
    for (size_t i = 0; i < m && !r.rc; ++i) r.digest(fill ? leaf_siblings_hashes + i * 4 * lfe : nullptr, lfe, compress != 0, val, i);
    const size_t m2 = r.count(8);
    if (!r.rc && m2 != m) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: %zu sibling hashes but %zu prefix lengths", m, m2);
    for (size_t i = 0; i < m2 && !r.rc; ++i) {
        const uint64_t v = r.u64();
        if (fill) prefix_lengths[i] = v;
    }
    const size_t m3 = r.count(8);
    if (!r.rc && m3 != m) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: %zu sibling hashes but %zu suffixes", m, m3);
    size_t at = 0;
    for (size_t i = 0; i < m3 && !r.rc; ++i) {
        const size_t k = r.count(per);
        if (fill) suffix_lengths[i] = k;
        if (!r.rc && fill && at + k > suffix_cap) return fail(AKP_ERR_BAD_LENGTH, "akp_deserialize_multipath: more than %zu suffix digests", suffix_cap);
        if (!r.rc && fill && k && !suffixes) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: suffixes is NULL");
        for (size_t j = 0; j < k && !r.rc; ++j, ++at) r.digest(fill ? suffixes + at * 4 * fe : nullptr, fe, compress != 0, val, at);
    }
    const size_t m4 = r.count(8);
    if (!r.rc && m4 != m) return fail(AKP_ERR_BAD_PARAMS, "akp_deserialize_multipath: %zu sibling hashes but %zu leaf indexes", m, m4);
    for (size_t i = 0; i < m4 && !r.rc; ++i) {
        const uint64_t v = r.u64();
        if (fill) leaf_indexes[i] = v;
    }
    if (int32_t rc = r.done()) return rc;
    if (m_out) *m_out = m;
    if (n_suffix_out) *n_suffix_out = at;
    return AKP_OK;
}
