// poseidon_opt.hpp -- host-side derivation of the "sparse partial rounds" form of a Poseidon instance.
//
// The reference applies the dense t x t MDS matrix in every round (sponge/poseidon/mod.rs:85-121).  In a partial
// round only lane 0 passes through the S-box, so the linear layers of consecutive partial rounds can be
// re-associated (Poseidon paper, appendix on optimised implementations).  The result is the SAME function on field
// elements -- outputs are identical canonical integers -- with t + (t-1) instead of t^2 constant products per
// partial round and one round-key addition instead of t.
//
// Derivation used here (x = state, M = MDS, S_0 = S-box on lane 0, c^(j) = round keys of partial round j = 1..RP):
//   every matrix A with invertible lower-right block A^ factors as  A = A'' * A',
//       A' = diag(1, A^),   A'' = [[a00, v^T A^-1], [w, I]]        (a00, v, w, A^ the blocks of A),
//   and A' commutes with S_0.  Going backwards from A_RP = M:   A_j = A''_j A'_j,   A_{j-1} = A'_j M.
//   With y_j = A'_j x_{j-1}:   y_{j+1} = A''_j S_0(y_j + A'_j c^(j)),  y_{RP+1} = x_RP, and the full round that
//   precedes the partial block applies M_pre = A_0 = A'_1 M instead of M.
//   Round keys of lanes 1.. are pushed forward through the (linear) lanes: only q_j[0] is added in round j and
//   the residue d is folded into the keys of the first full round after the block.
// Layout of `sparse` per partial round (2t entries): [ q0, a00, u_1..u_{t-1} (first row), w_1..w_{t-1} (first column) ].
#pragma once
#include <vector>

#include "fr.hpp"

namespace akp {

struct PoseidonOpt {
    bool ok = false;
    std::vector<Fr> ark_mod;  // [R][t]: original keys; row (half + RP) has the residue folded in
    std::vector<Fr> mpre;     // [t][t]
    std::vector<Fr> sparse;   // [RP][2t]
    bool scaled = false;      // poseidon_rescale_sparse applied: a00 == 1 in all partial rounds but the last
};

namespace optdetail {
typedef std::vector<Fr> Mat;  // row-major n x n
inline Mat matmul(const Mat& a, const Mat& b, unsigned n) {
    Mat r((size_t)n * n, fr_zero());
    for (unsigned i = 0; i < n; ++i)
        for (unsigned j = 0; j < n; ++j) {
            Fr acc = fr_zero();
            for (unsigned k = 0; k < n; ++k) acc = fr_add(acc, fr_mul(a[(size_t)i * n + k], b[(size_t)k * n + j]));
            r[(size_t)i * n + j] = acc;
        }
    return r;
}
// Gauss-Jordan inverse; returns false if singular
inline bool matinv(const Mat& a, unsigned n, Mat& out) {
    Mat m = a;
    out.assign((size_t)n * n, fr_zero());
    for (unsigned i = 0; i < n; ++i) out[(size_t)i * n + i] = fr_one();
    for (unsigned col = 0; col < n; ++col) {
        unsigned piv = col;
        while (piv < n && fr_is_zero(m[(size_t)piv * n + col])) ++piv;
        if (piv == n) return false;
        if (piv != col)
            for (unsigned j = 0; j < n; ++j) {
                std::swap(m[(size_t)piv * n + j], m[(size_t)col * n + j]);
                std::swap(out[(size_t)piv * n + j], out[(size_t)col * n + j]);
            }
        const Fr inv = fr_inv(m[(size_t)col * n + col]);
        for (unsigned j = 0; j < n; ++j) {
            m[(size_t)col * n + j] = fr_mul(m[(size_t)col * n + j], inv);
            out[(size_t)col * n + j] = fr_mul(out[(size_t)col * n + j], inv);
        }
        for (unsigned i = 0; i < n; ++i) {
            if (i == col) continue;
            const Fr f = m[(size_t)i * n + col];
            if (fr_is_zero(f)) continue;
            for (unsigned j = 0; j < n; ++j) {
                m[(size_t)i * n + j] = fr_sub(m[(size_t)i * n + j], fr_mul(f, m[(size_t)col * n + j]));
                out[(size_t)i * n + j] = fr_sub(out[(size_t)i * n + j], fr_mul(f, out[(size_t)col * n + j]));
            }
        }
    }
    return true;
}
}  // namespace optdetail

inline PoseidonOpt poseidon_optimize(unsigned t, unsigned full_rounds, unsigned partial_rounds, const std::vector<Fr>& ark,
                                     const std::vector<Fr>& mds) {
    using namespace optdetail;
    PoseidonOpt o;
    const unsigned half = full_rounds / 2, RP = partial_rounds, n1 = t - 1;
    if (t < 2 || half == 0 || RP == 0) return o;  // needs a full round on each side of the partial block
    struct Fac { Fr a00; std::vector<Fr> u, w; Mat hat; };
    std::vector<Fac> fac(RP + 1);  // 1-based
    Mat A = mds;
    for (unsigned j = RP; j >= 1; --j) {
        Fac f;
        f.a00 = A[0];
        std::vector<Fr> v(n1);
        f.w.resize(n1);
        f.hat.assign((size_t)n1 * n1, fr_zero());
        for (unsigned i = 0; i < n1; ++i) {
            v[i] = A[1 + i];
            f.w[i] = A[(size_t)(1 + i) * t];
            for (unsigned k = 0; k < n1; ++k) f.hat[(size_t)i * n1 + k] = A[(size_t)(1 + i) * t + 1 + k];
        }
        Mat hinv;
        if (!matinv(f.hat, n1, hinv)) return o;  // singular block: keep the dense form
        f.u.assign(n1, fr_zero());               // u^T = v^T * hat^-1
        for (unsigned k = 0; k < n1; ++k) {
            Fr acc = fr_zero();
            for (unsigned i = 0; i < n1; ++i) acc = fr_add(acc, fr_mul(v[i], hinv[(size_t)i * n1 + k]));
            f.u[k] = acc;
        }
        // A_{j-1} = A'_j * M,  A'_j = diag(1, hat)
        Mat Ap((size_t)t * t, fr_zero());
        Ap[0] = fr_one();
        for (unsigned i = 0; i < n1; ++i)
            for (unsigned k = 0; k < n1; ++k) Ap[(size_t)(1 + i) * t + 1 + k] = f.hat[(size_t)i * n1 + k];
        A = matmul(Ap, mds, t);
        fac[j] = std::move(f);
    }
    o.mpre = A;
    o.ark_mod = ark;
    o.sparse.assign((size_t)RP * 2 * t, fr_zero());
    std::vector<Fr> d(t, fr_zero());  // pending constant vector (d[0] is always folded immediately)
    for (unsigned j = 1; j <= RP; ++j) {
        const Fac& f = fac[j];
        const Fr* c = &ark[(size_t)(half + j - 1) * t];
        // k_j = A'_j c^(j):  k[0] = c[0],  k[1..] = hat * c[1..];   q = d + k
        std::vector<Fr> q(t);
        q[0] = fr_add(d[0], c[0]);
        for (unsigned i = 0; i < n1; ++i) {
            Fr acc = fr_zero();
            for (unsigned k = 0; k < n1; ++k) acc = fr_add(acc, fr_mul(f.hat[(size_t)i * n1 + k], c[1 + k]));
            q[1 + i] = fr_add(d[1 + i], acc);
        }
        Fr* s = &o.sparse[(size_t)(j - 1) * 2 * t];
        s[0] = q[0];
        s[1] = f.a00;
        for (unsigned i = 0; i < n1; ++i) {
            s[2 + i] = f.u[i];
            s[2 + n1 + i] = f.w[i];
        }
        // d_{j+1} = A''_j [0; q_rest]:  d[0] = u . q_rest,  d[1..] = q_rest
        Fr acc = fr_zero();
        for (unsigned i = 0; i < n1; ++i) acc = fr_add(acc, fr_mul(f.u[i], q[1 + i]));
        d[0] = acc;
        for (unsigned i = 0; i < n1; ++i) d[1 + i] = q[1 + i];
    }
    Fr* nxt = &o.ark_mod[(size_t)(half + RP) * t];  // first full round after the block
    for (unsigned i = 0; i < t; ++i) nxt[i] = fr_add(nxt[i], d[i]);
    o.ok = true;
    return o;
}

// Remove the multiplication a00 * S(x) from the sparse partial rounds.  Lane 0 is carried scaled by a per-round constant
// d_j (d_0 = 1): the kernel holds lambda_j = d_j * lane0_j, adds q0'_j = d_j * q0_j and raises to alpha, which gives
// T_j = d_j^alpha * s_j.  With d_{j+1} = d_j^alpha / a00_j the next scaled lane is simply
//     lambda_{j+1} = T_j + sum_i (d_{j+1} u_i) y_i          (no product on the S-box output),
// the other lanes take y_i += (w_i / d_j^alpha) T_j, and the last partial round returns to the true lane with
// a00' = a00 / d_j^alpha.  Pure re-parameterisation: every state the full rounds see is unchanged.
// Needs a00_j != 0 for all but the last round; otherwise the constants are left as they are (scaled stays false).
inline void poseidon_rescale_sparse(PoseidonOpt& o, unsigned t, unsigned partial_rounds, uint64_t alpha) {
    if (!o.ok || partial_rounds < 2) return;
    const unsigned n1 = t - 1;
    for (unsigned j = 0; j + 1 < partial_rounds; ++j)
        if (fr_is_zero(o.sparse[(size_t)j * 2 * t + 1])) return;
    Fr d = fr_one();
    for (unsigned j = 0; j < partial_rounds; ++j) {
        Fr* s = &o.sparse[(size_t)j * 2 * t];
        const Fr e = fr_pow_small(d, alpha), einv = fr_inv(e);
        s[0] = fr_mul(s[0], d);
        for (unsigned i = 0; i < n1; ++i) s[2 + n1 + i] = fr_mul(s[2 + n1 + i], einv);
        if (j + 1 < partial_rounds) {
            const Fr dn = fr_mul(e, fr_inv(s[1]));
            s[1] = fr_one();  // kernels that still multiply by the a00 slot stay correct
            for (unsigned i = 0; i < n1; ++i) s[2 + i] = fr_mul(s[2 + i], dn);
            d = dn;
        } else {
            s[1] = fr_mul(s[1], einv);
        }
    }
    o.scaled = true;
}

// Second re-parameterisation, for kernels whose cost is instruction count rather than the latency of lane 0: make the
// FIRST-COLUMN coefficient of lane 1 equal to 1, so that lane 1 takes the S-box output with a plain addition
// (a whole field product less per partial round; row 0 keeps its three products).  Lane 0 is carried as
// lambda_j = d_j * lane0_j with d_j = w_{1,j}^(1/alpha)  =>  T_j = (lambda_j + d_j q0_j)^alpha = w_{1,j} s_j, hence
//     y_1 += T_j,    y_i += (w_{i,j} / w_{1,j}) T_j  (i >= 2),
//     lambda_{j+1} = (d_{j+1} a00_j / w_{1,j}) T_j + sum_i (d_{j+1} u_{i,j}) y_i      (last round: d_{j+1} := 1),
// and row 0 of M_pre is multiplied by d_0.  The d_j are independent of each other (no recursion).  Requires
// gcd(alpha, p - 1) = 1 (alpha-th roots) and w_{1,j} != 0; otherwise returns false and leaves `o` untouched.
namespace optdetail {
// x^e, e given as little-endian 64-bit words
inline Fr fr_pow_words(const Fr& x, const uint64_t* e, int words) {
    Fr r = fr_one();
    for (int i = 64 * words - 1; i >= 0; --i) {
        r = fr_sqr(r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fr_mul(r, x);
    }
    return r;
}
// alpha^-1 mod (p - 1) as 4 words, for 2 <= alpha < 2^16 with gcd(alpha, p - 1) = 1:  (1 + k (p - 1)) / alpha for the
// k < alpha that makes the division exact
inline bool inv_alpha_mod_pm1(uint64_t alpha, uint64_t out[4]) {
    if (alpha < 2 || alpha >= 65536) return false;
    const uint64_t pm1[4] = {((uint64_t)AKP_P1 << 32) | (uint64_t)(AKP_P0 - 1u), ((uint64_t)AKP_P3 << 32) | AKP_P2,
                             ((uint64_t)AKP_P5 << 32) | AKP_P4, ((uint64_t)AKP_P7 << 32) | AKP_P6};
    for (uint64_t k = 1; k < alpha; ++k) {
        uint64_t v[5];
        unsigned __int128 c = 1;  // v = 1 + k * (p - 1)
        for (int i = 0; i < 4; ++i) {
            c += (unsigned __int128)pm1[i] * k;
            v[i] = (uint64_t)c;
            c >>= 64;
        }
        v[4] = (uint64_t)c;
        unsigned __int128 rem = 0;  // v / alpha, most significant word first
        uint64_t q[5];
        for (int i = 4; i >= 0; --i) {
            const unsigned __int128 cur = (rem << 64) | v[i];
            q[i] = (uint64_t)(cur / alpha);
            rem = cur % alpha;
        }
        if (rem == 0 && q[4] == 0) {
            for (int i = 0; i < 4; ++i) out[i] = q[i];
            return true;
        }
    }
    return false;
}
}  // namespace optdetail
inline bool poseidon_rescale_sparse_lane1(PoseidonOpt& o, unsigned t, unsigned partial_rounds, uint64_t alpha) {
    if (!o.ok || o.scaled || partial_rounds < 1 || t < 2) return false;
    const unsigned n1 = t - 1;
    uint64_t einv[4];
    if (!optdetail::inv_alpha_mod_pm1(alpha, einv)) return false;
    std::vector<Fr> d(partial_rounds + 1);
    for (unsigned j = 0; j < partial_rounds; ++j) {
        const Fr w1 = o.sparse[(size_t)j * 2 * t + 2 + n1];
        if (fr_is_zero(w1)) return false;
        d[j] = optdetail::fr_pow_words(w1, einv, 4);
        if (!fr_eq(fr_pow_small(d[j], alpha), w1)) return false;  // alpha-th root does not exist / is not this one
    }
    d[partial_rounds] = fr_one();
    for (unsigned j = 0; j < partial_rounds; ++j) {
        Fr* s = &o.sparse[(size_t)j * 2 * t];
        const Fr w1inv = fr_inv(s[2 + n1]);
        s[0] = fr_mul(s[0], d[j]);
        s[1] = fr_mul(fr_mul(s[1], w1inv), d[j + 1]);
        for (unsigned i = 0; i < n1; ++i) {
            s[2 + i] = fr_mul(s[2 + i], d[j + 1]);
            s[2 + n1 + i] = (i == 0) ? fr_one() : fr_mul(s[2 + n1 + i], w1inv);
        }
    }
    for (unsigned c = 0; c < t; ++c) o.mpre[c] = fr_mul(o.mpre[c], d[0]);  // row 0 of M_pre
    return true;
}

// Third form ("full form", register kernel for t = 3): the lane scaling of the lane-1 form is extended through the full
// rounds.  Lane i enters a full round multiplied by delta_i, its key is scaled likewise, the S-box output is
// T_i = delta_i^alpha s_i, and the output scale of row i is chosen as delta'_i = delta_i^alpha / M_ii so that the
// DIAGONAL coefficient of the (per-round) linear layer is 1:  row i = T_i + sum_{j != i} F_ij T_j  -- one product
// less per row.  Exception: row 0 of the round before the partial block must deliver lane 0 with the scale the
// lane-1 form prescribes (d_0), and the last round delivers every lane with the scale 2^-5.  The lanes also ENTER with
// the scale 2^-5: x * 2^261 * 2^-5 = x * 2^256 is the wire value itself, so the kernel's conversions from and to the
// wire format are a re-limbing and a canonicalisation, with no field product.  The last partial round uses its free
// output scale to make a00 = 1.  Lanes 1.. run through the partial block with the constant scales g_i they got from the last
// full round before it.  Input: the plain sparse form (poseidon_optimize).  Output: keys [R][t] (partial-round rows
// unused), one matrix per full round [RF][t][t], sparse [RP][2t].  Returns false (outputs untouched) when a required
// coefficient is zero or alpha-th roots do not exist.
struct PoseidonFullForm {
    std::vector<Fr> ark, fmats, sparse;
};
inline bool poseidon_full_form(const PoseidonOpt& o, unsigned t, unsigned full_rounds, unsigned partial_rounds, uint64_t alpha,
                               const std::vector<Fr>& mds, PoseidonFullForm& out) {
    if (!o.ok || o.scaled || t < 2 || partial_rounds < 1) return false;
    const unsigned half = full_rounds / 2, RP = partial_rounds, R = full_rounds + RP, n1 = t - 1;
    uint64_t einv[4];
    if (!optdetail::inv_alpha_mod_pm1(alpha, einv)) return false;
    PoseidonFullForm f;
    f.ark = o.ark_mod;
    f.fmats.assign((size_t)full_rounds * t * t, fr_zero());
    f.sparse = o.sparse;
    const Fr inv32 = fr_inv(fr_to_mont(Fr{{32u, 0, 0, 0, 0, 0, 0, 0}}));
    std::vector<Fr> delta(t, inv32), e(t), dn(t);
    auto sp = [&](unsigned j, unsigned k) -> const Fr& { return o.sparse[(size_t)j * 2 * t + k]; };
    // d_j = (g_1 w_{1,j})^(1/alpha); g_1 is known once the round before the block has chosen its row scales
    auto root = [&](const Fr& v, Fr& r) {
        r = optdetail::fr_pow_words(v, einv, 4);
        return fr_eq(fr_pow_small(r, alpha), v);
    };
    auto full_round = [&](unsigned r, unsigned fr_index, const std::vector<Fr>& mat, bool last, const Fr* row0_scale) {
        for (unsigned i = 0; i < t; ++i) {
            f.ark[(size_t)r * t + i] = fr_mul(delta[i], o.ark_mod[(size_t)r * t + i]);
            e[i] = fr_pow_small(delta[i], alpha);
        }
        for (unsigned i = 0; i < t; ++i) {
            if (last) dn[i] = inv32;
            else if (i == 0 && row0_scale) dn[i] = *row0_scale;
            else {
                if (fr_is_zero(mat[(size_t)i * t + i])) return false;
                dn[i] = fr_mul(e[i], fr_inv(mat[(size_t)i * t + i]));
            }
        }
        for (unsigned i = 0; i < t; ++i)
            for (unsigned j = 0; j < t; ++j)
                f.fmats[((size_t)fr_index * t + i) * t + j] = fr_mul(fr_mul(dn[i], mat[(size_t)i * t + j]), fr_inv(e[j]));
        return true;
    };
    for (unsigned r = 0; r + 1 < half; ++r) {
        if (!full_round(r, r, mds, false, nullptr)) return false;
        delta = dn;
    }
    // round before the block: rows 1.. first (their scales g_i define d_0), then row 0 with the prescribed scale
    {
        const unsigned r = half - 1;
        std::vector<Fr> ee(t);
        for (unsigned i = 0; i < t; ++i) ee[i] = fr_pow_small(delta[i], alpha);
        if (fr_is_zero(o.mpre[(size_t)1 * t + 1]) || fr_is_zero(sp(0, 2 + n1))) return false;
        const Fr g1 = fr_mul(ee[1], fr_inv(o.mpre[(size_t)1 * t + 1]));
        Fr d0;
        if (!root(fr_mul(g1, sp(0, 2 + n1)), d0)) return false;
        if (!full_round(r, r, o.mpre, false, &d0)) return false;
        delta = dn;
    }
    std::vector<Fr> g(delta);  // g[i], i >= 1: lane scales through the block
    std::vector<Fr> d(RP + 1);
    for (unsigned j = 0; j < RP; ++j) {
        if (fr_is_zero(sp(j, 2 + n1))) return false;
        if (!root(fr_mul(g[1], sp(j, 2 + n1)), d[j])) return false;
    }
    if (!fr_eq(d[0], delta[0])) return false;
    if (fr_is_zero(sp(RP - 1, 1))) return false;
    d[RP] = fr_mul(fr_mul(g[1], sp(RP - 1, 2 + n1)), fr_inv(sp(RP - 1, 1)));  // makes a00 of the last partial round 1
    for (unsigned j = 0; j < RP; ++j) {
        Fr* s = &f.sparse[(size_t)j * 2 * t];
        const Fr gw1inv = fr_inv(fr_mul(g[1], sp(j, 2 + n1)));
        s[0] = fr_mul(d[j], sp(j, 0));
        s[1] = fr_mul(fr_mul(d[j + 1], sp(j, 1)), gw1inv);  // == 1 in the last round
        for (unsigned i = 0; i < n1; ++i) {
            s[2 + i] = fr_mul(fr_mul(d[j + 1], sp(j, 2 + i)), fr_inv(g[1 + i]));
            s[2 + n1 + i] = (i == 0) ? fr_one() : fr_mul(fr_mul(g[1 + i], sp(j, 2 + n1 + i)), gw1inv);
        }
    }
    delta[0] = d[RP];
    for (unsigned r = half + RP; r < R; ++r) {
        if (!full_round(r, r - RP, mds, r + 1 == R, nullptr)) return false;
        delta = dn;
    }
    out = std::move(f);
    return true;
}

// (round-0 key of lane i)^alpha for the round keys the kernels actually use: what the first S-box of a lane that
// enters the permutation as zero produces (PoseidonConsts::sbox0)
inline std::vector<Fr> poseidon_sbox0(const std::vector<Fr>& ark_used, uint32_t t, uint64_t alpha) {
    std::vector<Fr> out(t);
    for (uint32_t i = 0; i < t; ++i) out[i] = fr_pow_small(ark_used[i], alpha);
    return out;
}

}  // namespace akp
