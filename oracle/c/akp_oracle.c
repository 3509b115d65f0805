/*
 * akp_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (crypto_primitives_amd/csrc) never links or calls it.
 *
 * "Reference-shaped": it performs the same algorithmic steps as the Rust reference
 * (dense MDS every round, MSB-first square-and-multiply S-box, a fresh sponge per
 * hash, one hash per Merkle node, barrier per level, bit-by-bit conditional point
 * additions), so it doubles as the timed CPU baseline ("kind": "port").
 *
 * Reference lines followed (relative to /root/reference/crypto-primitives/src):
 *   Poseidon permutation   sponge/poseidon/mod.rs:66-121
 *   duplex sponge          sponge/poseidon/mod.rs:124-186,223-257,324-344
 *   Poseidon CRH / 2-to-1  crh/poseidon/mod.rs:30-79
 *   Pedersen CRH           crh/pedersen/mod.rs:76-129,158-209
 *   Bowe-Hopwood CRH       crh/bowe_hopwood/mod.rs:114-186,202-239
 *   MerkleTree::new        merkle_tree/mod.rs:411-523
 * The field / curve arithmetic lives in ark-ff / ark-ec (un-vendored git deps,
 * /root/reference/Cargo.toml:46-58) and is restated from the published definitions:
 * 4x64-bit Montgomery (R = 2^256) over BLS12-381 Fr; Jubjub twisted Edwards a=-1.
 *
 * Pinning: Poseidon functions are pinned by the reference KATs through
 * tests/test_oracle_c.py (C == python oracle == reference constants).  Pedersen /
 * Bowe-Hopwood / Merkle digests are PARITY UNPINNED at value level as far as the reference goes (no absolute vectors
 * exist in the reference); they are checked against the python big-int oracle and
 * structural identities.
 *
 * Wire format everywhere: Fr = 4 x u64 little-endian limbs, Montgomery form, reduced.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr;

static const fr FR_P = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const fr FR_R = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};  /* R mod p = mont(1) */
static const fr FR_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}}; /* R^2 mod p */
#define FR_INV 0xfffffffeffffffffULL /* -p^{-1} mod 2^64 */

static inline int fr_geq_p(const fr *a) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > FR_P.l[i]) return 1;
        if (a->l[i] < FR_P.l[i]) return 0;
    }
    return 1;
}
static inline void fr_sub_p(fr *a) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - FR_P.l[i] - (uint64_t)b;
        a->l[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline void fr_add(fr *r, const fr *a, const fr *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        r->l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (fr_geq_p(r)) fr_sub_p(r); /* p < 2^255 so no carry out */
}
static inline void fr_sub(fr *r, const fr *a, const fr *b) {
    u128 bw = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)bw;
        t[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + FR_P.l[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}
static inline void fr_neg(fr *r, const fr *a) {
    fr z = {{0, 0, 0, 0}};
    fr_sub(r, &z, a);
}
static inline int fr_is_zero(const fr *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_eq(const fr *a, const fr *b) { return memcmp(a, b, sizeof(fr)) == 0; }

/* CIOS Montgomery multiplication */
static inline void fr_mul(fr *r, const fr *a, const fr *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_INV;
        c = (u128)m * FR_P.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * FR_P.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq_p(&o)) fr_sub_p(&o);
    *r = o;
}
static inline void fr_sqr(fr *r, const fr *a) { fr_mul(r, a, a); }

/* ark-ff Field::pow: MSB-first square-and-multiply without leading zeros */
static void fr_pow_u64(fr *r, const fr *a, uint64_t e) {
    fr res = FR_R;
    int started = 0;
    for (int i = 63; i >= 0; --i) {
        int bit = (e >> i) & 1;
        if (!started && !bit) continue;
        started = 1;
        fr_sqr(&res, &res);
        if (bit) fr_mul(&res, &res, a);
    }
    *r = res;
}
static void fr_inv(fr *r, const fr *a) { /* a^(p-2) */
    uint64_t e[4] = {FR_P.l[0] - 2, FR_P.l[1], FR_P.l[2], FR_P.l[3]};
    fr res = FR_R;
    for (int i = 255; i >= 0; --i) {
        fr_sqr(&res, &res);
        if ((e[i >> 6] >> (i & 63)) & 1) fr_mul(&res, &res, a);
    }
    *r = res;
}
static void fr_from_mont(fr *r, const fr *a) {
    fr one = {{1, 0, 0, 0}};
    fr_mul(r, a, &one);
}

/* ------------------------------------------------------------------ exported field ops */
void orc_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { fr_mul((fr *)out, (const fr *)a, (const fr *)b); }
void orc_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out) { fr_add((fr *)out, (const fr *)a, (const fr *)b); }
void orc_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out) { fr_sub((fr *)out, (const fr *)a, (const fr *)b); }
void orc_fr_inv(const uint64_t *a, uint64_t *out) { fr_inv((fr *)out, (const fr *)a); }
void orc_fr_to_mont(const uint64_t *canon, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) fr_mul((fr *)(out + 4 * i), (const fr *)(canon + 4 * i), &FR_R2);
}
void orc_fr_from_mont(const uint64_t *mont, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) fr_from_mont((fr *)(out + 4 * i), (const fr *)(mont + 4 * i));
}

/* ------------------------------------------------------------------ Poseidon */
#define ORC_MAX_T 16
typedef struct {
    uint32_t full_rounds, partial_rounds, rate, capacity;
    uint64_t alpha;
    fr *ark; /* [(RF+RP)][t] */
    fr *mds; /* [t][t] */
} orc_poseidon;

void *orc_poseidon_new(uint32_t full_rounds, uint32_t partial_rounds, uint64_t alpha, uint32_t rate, uint32_t capacity,
                       const uint64_t *ark_mont, const uint64_t *mds_mont) {
    uint32_t t = rate + capacity;
    if (t > ORC_MAX_T || t == 0) return NULL;
    orc_poseidon *p = (orc_poseidon *)malloc(sizeof *p);
    p->full_rounds = full_rounds; p->partial_rounds = partial_rounds; p->rate = rate; p->capacity = capacity; p->alpha = alpha;
    size_t na = (size_t)(full_rounds + partial_rounds) * t, nm = (size_t)t * t;
    p->ark = (fr *)malloc(na * sizeof(fr)); p->mds = (fr *)malloc(nm * sizeof(fr));
    memcpy(p->ark, ark_mont, na * sizeof(fr)); memcpy(p->mds, mds_mont, nm * sizeof(fr));
    return p;
}
void orc_poseidon_free(void *h) {
    orc_poseidon *p = (orc_poseidon *)h;
    if (!p) return;
    free(p->ark); free(p->mds); free(p);
}

/* sponge/poseidon/mod.rs:98-121 */
static void poseidon_permute(const orc_poseidon *p, fr *state) {
    const uint32_t t = p->rate + p->capacity, half = p->full_rounds / 2, total = p->full_rounds + p->partial_rounds;
    fr ns[ORC_MAX_T];
    for (uint32_t r = 0; r < total; ++r) {
        for (uint32_t i = 0; i < t; ++i) fr_add(&state[i], &state[i], &p->ark[(size_t)r * t + i]); /* apply_ark :79-83 */
        if (r < half || r >= half + p->partial_rounds) {                                          /* apply_s_box :66-77 */
            for (uint32_t i = 0; i < t; ++i) fr_pow_u64(&state[i], &state[i], p->alpha);
        } else {
            fr_pow_u64(&state[0], &state[0], p->alpha);
        }
        for (uint32_t i = 0; i < t; ++i) { /* apply_mds :85-96 */
            fr cur = {{0, 0, 0, 0}}, term;
            for (uint32_t j = 0; j < t; ++j) {
                fr_mul(&term, &state[j], &p->mds[(size_t)i * t + j]);
                fr_add(&cur, &cur, &term);
            }
            ns[i] = cur;
        }
        memcpy(state, ns, t * sizeof(fr));
    }
}

/* duplex sponge restricted to: absorb one slice from a fresh sponge, squeeze n_out.
 * sponge/poseidon/mod.rs:124-153 (absorb_internal), :236-257, :324-344, :156-186 */
typedef struct { fr state[ORC_MAX_T]; int squeezing; uint32_t idx; } sponge_t;

static void sponge_new(const orc_poseidon *p, sponge_t *s) {
    memset(s->state, 0, sizeof s->state);
    s->squeezing = 0; s->idx = 0;
    (void)p;
}
static void sponge_absorb_internal(const orc_poseidon *p, sponge_t *s, uint32_t idx, const fr *e, size_t n) {
    for (;;) {
        if (idx + n <= p->rate) {
            for (size_t i = 0; i < n; ++i) fr_add(&s->state[p->capacity + i + idx], &s->state[p->capacity + i + idx], &e[i]);
            s->squeezing = 0; s->idx = idx + (uint32_t)n;
            return;
        }
        uint32_t k = p->rate - idx;
        for (uint32_t i = 0; i < k; ++i) fr_add(&s->state[p->capacity + i + idx], &s->state[p->capacity + i + idx], &e[i]);
        poseidon_permute(p, s->state);
        e += k; n -= k; idx = 0;
    }
}
static void sponge_absorb(const orc_poseidon *p, sponge_t *s, const fr *e, size_t n) {
    if (n == 0) return;
    if (!s->squeezing) {
        uint32_t idx = s->idx;
        if (idx == p->rate) { poseidon_permute(p, s->state); idx = 0; }
        sponge_absorb_internal(p, s, idx, e, n);
    } else {
        sponge_absorb_internal(p, s, 0, e, n);
    }
}
static void sponge_squeeze_internal(const orc_poseidon *p, sponge_t *s, uint32_t idx, fr *out, size_t n) {
    for (;;) {
        if (idx + n <= p->rate) {
            memcpy(out, &s->state[p->capacity + idx], n * sizeof(fr));
            s->squeezing = 1; s->idx = idx + (uint32_t)n;
            return;
        }
        uint32_t k = p->rate - idx;
        memcpy(out, &s->state[p->capacity + idx], k * sizeof(fr));
        out += k; n -= k;
        if (n != 0) poseidon_permute(p, s->state);
        idx = 0;
    }
}
static void sponge_squeeze(const orc_poseidon *p, sponge_t *s, fr *out, size_t n) {
    if (!s->squeezing) {
        poseidon_permute(p, s->state);
        sponge_squeeze_internal(p, s, 0, out, n);
    } else {
        uint32_t idx = s->idx;
        if (idx == p->rate) { poseidon_permute(p, s->state); idx = 0; }
        sponge_squeeze_internal(p, s, idx, out, n);
    }
}
/* crh/poseidon/mod.rs:30-40 */
static void poseidon_crh(const orc_poseidon *p, const fr *in, size_t n, fr *out) {
    sponge_t s; sponge_new(p, &s);
    sponge_absorb(p, &s, in, n);
    sponge_squeeze(p, &s, out, 1);
}
/* crh/poseidon/mod.rs:66-79 */
static void poseidon_two_to_one(const orc_poseidon *p, const fr *l, const fr *r, fr *out) {
    sponge_t s; sponge_new(p, &s);
    sponge_absorb(p, &s, l, 1);
    sponge_absorb(p, &s, r, 1);
    sponge_squeeze(p, &s, out, 1);
}

/* generic sponge transcript for tests: ops[k] > 0 => absorb ops[k] elements (taken in order
 * from `in`), ops[k] < 0 => squeeze -ops[k] elements (appended to out). */
void orc_poseidon_sponge_script(void *h, const int32_t *ops, size_t n_ops, const uint64_t *in, uint64_t *out) {
    const orc_poseidon *p = (const orc_poseidon *)h;
    sponge_t s; sponge_new(p, &s);
    const fr *ip = (const fr *)in; fr *op = (fr *)out;
    for (size_t k = 0; k < n_ops; ++k) {
        if (ops[k] > 0) { sponge_absorb(p, &s, ip, (size_t)ops[k]); ip += ops[k]; }
        else if (ops[k] < 0) { sponge_squeeze(p, &s, op, (size_t)(-ops[k])); op += -ops[k]; }
    }
}

/* ------------------------------------------------------------------ thread pool helper */
typedef void (*range_fn)(void *ctx, size_t lo, size_t hi);
typedef struct { range_fn fn; void *ctx; size_t lo, hi; } job_t;
static void *job_main(void *a) { job_t *j = (job_t *)a; j->fn(j->ctx, j->lo, j->hi); return NULL; }
static void parallel_for(size_t n, int threads, range_fn fn, void *ctx) {
    if (threads <= 1 || n < 2) { fn(ctx, 0, n); return; }
    if ((size_t)threads > n) threads = (int)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t].fn = fn; jobs[t].ctx = ctx;
        jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
        pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------ Poseidon batches */
typedef struct { const orc_poseidon *p; fr *a; const fr *b; const fr *c; fr *out; size_t k; } pb_ctx;
static void permute_range(void *c, size_t lo, size_t hi) {
    pb_ctx *x = (pb_ctx *)c; uint32_t t = x->p->rate + x->p->capacity;
    for (size_t i = lo; i < hi; ++i) poseidon_permute(x->p, x->a + i * t);
}
void orc_poseidon_permute_batch(void *h, uint64_t *states, size_t n, int threads) {
    pb_ctx c = {(const orc_poseidon *)h, (fr *)states, NULL, NULL, NULL, 0};
    parallel_for(n, threads, permute_range, &c);
}
static void crh_range(void *c, size_t lo, size_t hi) {
    pb_ctx *x = (pb_ctx *)c;
    for (size_t i = lo; i < hi; ++i) poseidon_crh(x->p, x->b + i * x->k, x->k, x->out + i);
}
void orc_poseidon_crh_batch(void *h, const uint64_t *inputs, size_t n, size_t elems_per_input, uint64_t *out, int threads) {
    pb_ctx c = {(const orc_poseidon *)h, NULL, (const fr *)inputs, NULL, (fr *)out, elems_per_input};
    parallel_for(n, threads, crh_range, &c);
}
static void t21_range(void *c, size_t lo, size_t hi) {
    pb_ctx *x = (pb_ctx *)c;
    for (size_t i = lo; i < hi; ++i) poseidon_two_to_one(x->p, x->b + i, x->c + i, x->out + i);
}
void orc_poseidon_two_to_one_batch(void *h, const uint64_t *left, const uint64_t *right, size_t n, uint64_t *out, int threads) {
    pb_ctx c = {(const orc_poseidon *)h, NULL, (const fr *)left, (const fr *)right, (fr *)out, 0};
    parallel_for(n, threads, t21_range, &c);
}

/* MerkleTree::new, Poseidon leaf CRH + Poseidon 2-to-1, IdentityDigestConverter
 * (merkle_tree/mod.rs:411-523; config shape merkle_tree/tests/mod.rs:198-206) */
typedef struct { const orc_poseidon *p; const fr *child; fr *nodes; size_t first; } lvl_ctx;
static void level_range(void *c, size_t lo, size_t hi) {
    lvl_ctx *x = (lvl_ctx *)c;
    for (size_t i = lo; i < hi; ++i) poseidon_two_to_one(x->p, &x->child[2 * i], &x->child[2 * i + 1], &x->nodes[x->first + i]);
}
int orc_poseidon_merkle_build(void *leaf_h, void *two_h, const uint64_t *leaves, size_t n_leaves, size_t leaf_len,
                              uint64_t *leaf_nodes, uint64_t *non_leaf_nodes, int threads) {
    if (n_leaves < 2 || (n_leaves & (n_leaves - 1))) return 1; /* :430-433 */
    orc_poseidon_crh_batch(leaf_h, leaves, n_leaves, leaf_len, leaf_nodes, threads);
    fr *nl = (fr *)non_leaf_nodes;
    size_t width = n_leaves / 2, first = width - 1; /* bottom non-leaf level starts at n/2-1 */
    lvl_ctx c = {(const orc_poseidon *)two_h, (const fr *)leaf_nodes, nl, first};
    parallel_for(width, threads, level_range, &c);
    while (width > 1) { /* upper levels, barrier per level (:486-515) */
        size_t child_first = first;
        width /= 2; first = width - 1;
        lvl_ctx u = {(const orc_poseidon *)two_h, nl + child_first, nl, first};
        parallel_for(width, threads, level_range, &u);
    }
    return 0;
}

/* ------------------------------------------------------------------ Jubjub (twisted Edwards a = -1) */
/* d = -(10240/10241) mod p, Montgomery form */
static fr TE_D, TE_2D;
static int te_init_done = 0;
static void te_init(void) {
    if (te_init_done) return;
    /* canonical d = 19257038036680949359750312669786877991949435402254120286184196891950884077233 */
    fr dc = {{0x01065fd6d6343eb1ULL, 0x292d7f6d37579d26ULL, 0xf5fd9207e6bd7fd4ULL, 0x2a9318e74bfa2b48ULL}};
    fr_mul(&TE_D, &dc, &FR_R2);
    fr_add(&TE_2D, &TE_D, &TE_D);
    te_init_done = 1;
}
typedef struct { fr x, y, z, t; } tep; /* extended: x=X/Z, y=Y/Z, T=XY/Z (ark-ec te Projective) */

static void te_identity(tep *p) { memset(p, 0, sizeof *p); p->y = FR_R; p->z = FR_R; }
static void te_from_affine(tep *p, const fr *x, const fr *y) { p->x = *x; p->y = *y; p->z = FR_R; fr_mul(&p->t, x, y); }
/* unified complete addition (add-2008-hwcd, a = -1) */
static void te_add(tep *r, const tep *p, const tep *q) {
    fr a, b, c, d, e, f, g, h, t0, t1;
    fr_mul(&a, &p->x, &q->x);
    fr_mul(&b, &p->y, &q->y);
    fr_mul(&c, &p->t, &q->t); fr_mul(&c, &c, &TE_D);
    fr_mul(&d, &p->z, &q->z);
    fr_add(&t0, &p->x, &p->y); fr_add(&t1, &q->x, &q->y);
    fr_mul(&e, &t0, &t1); fr_sub(&e, &e, &a); fr_sub(&e, &e, &b);
    fr_sub(&f, &d, &c);
    fr_add(&g, &d, &c);
    fr_add(&h, &b, &a); /* b - a*A with A = -1 */
    fr_mul(&r->x, &e, &f);
    fr_mul(&r->y, &g, &h);
    fr_mul(&r->z, &f, &g);
    fr_mul(&r->t, &e, &h);
}
static void te_double(tep *r, const tep *p) { te_add(r, p, p); }
static void te_neg(tep *r, const tep *p) { *r = *p; fr_neg(&r->x, &p->x); fr_neg(&r->t, &p->t); }
static void te_to_affine(const tep *p, fr *x, fr *y) {
    fr zi; fr_inv(&zi, &p->z);
    fr_mul(x, &p->x, &zi); fr_mul(y, &p->y, &zi);
}

typedef struct { uint32_t window_size, num_windows; tep *gens; /* [N][W] */ } orc_curve_params;

void *orc_te_params_new(uint32_t window_size, uint32_t num_windows, const uint64_t *gens_affine_xy_mont) {
    te_init();
    orc_curve_params *p = (orc_curve_params *)malloc(sizeof *p);
    p->window_size = window_size; p->num_windows = num_windows;
    size_t n = (size_t)window_size * num_windows;
    p->gens = (tep *)malloc(n * sizeof(tep));
    for (size_t i = 0; i < n; ++i) te_from_affine(&p->gens[i], (const fr *)(gens_affine_xy_mont + 8 * i), (const fr *)(gens_affine_xy_mont + 8 * i + 4));
    return p;
}
void orc_te_params_free(void *h) { orc_curve_params *p = (orc_curve_params *)h; if (!p) return; free(p->gens); free(p); }

static inline int bit_at(const uint8_t *d, size_t i) { return (d[i >> 3] >> (i & 7)) & 1; } /* pedersen/mod.rs:200-209 */

/* crh/pedersen/mod.rs:76-129; msg_len*8 <= W*N (caller checks); returns affine x||y (Montgomery) */
static void pedersen_eval(const orc_curve_params *p, const uint8_t *msg, size_t msg_len, fr *out_xy) {
    size_t total_bits = (size_t)p->window_size * p->num_windows;
    size_t padded_len = total_bits / 8;
    size_t have_len = msg_len < padded_len ? padded_len : msg_len; /* zero-pad :91-99 */
    size_t nbits = have_len * 8;
    tep acc; te_identity(&acc);
    size_t n_chunks = (nbits + p->window_size - 1) / p->window_size;
    if (n_chunks > p->num_windows) n_chunks = p->num_windows;
    for (size_t i = 0; i < n_chunks; ++i) {
        tep enc; te_identity(&enc);
        for (uint32_t j = 0; j < p->window_size; ++j) {
            size_t bi = i * p->window_size + j;
            if (bi >= nbits) break;
            int bit = (bi < msg_len * 8) ? bit_at(msg, bi) : 0;
            if (bit) te_add(&enc, &enc, &p->gens[i * p->window_size + j]);
        }
        te_add(&acc, &acc, &enc);
    }
    te_to_affine(&acc, &out_xy[0], &out_xy[1]);
}
/* crh/bowe_hopwood/mod.rs:114-186; returns x (Montgomery) */
static void bh_eval(const orc_curve_params *p, const uint8_t *msg, size_t msg_len, fr *out_x) {
    size_t nbits = msg_len * 8;
    size_t padded = (nbits + 2) / 3 * 3; /* pad to multiple of 3 only :131-138 */
    size_t n_chunks_total = padded / 3;
    tep acc; te_identity(&acc);
    for (size_t s = 0; s < p->num_windows; ++s) {
        if (s * p->window_size >= n_chunks_total) break;
        tep seg; te_identity(&seg);
        for (uint32_t c = 0; c < p->window_size; ++c) {
            size_t ci = s * p->window_size + c;
            if (ci >= n_chunks_total) break;
            const tep *g = &p->gens[s * p->window_size + c];
            int b0 = (3 * ci < nbits) ? bit_at(msg, 3 * ci) : 0;
            int b1 = (3 * ci + 1 < nbits) ? bit_at(msg, 3 * ci + 1) : 0;
            int b2 = (3 * ci + 2 < nbits) ? bit_at(msg, 3 * ci + 2) : 0;
            tep enc = *g, dbl;
            if (b0) te_add(&enc, &enc, g);
            if (b1) { te_double(&dbl, g); te_add(&enc, &enc, &dbl); }
            if (b2) te_neg(&enc, &enc);
            te_add(&seg, &seg, &enc);
        }
        te_add(&acc, &acc, &seg);
    }
    fr y; te_to_affine(&acc, out_x, &y);
}

typedef struct { const orc_curve_params *p; const uint8_t *msgs; size_t msg_len; fr *out; int kind; } cb_ctx;
static void curve_range(void *c, size_t lo, size_t hi) {
    cb_ctx *x = (cb_ctx *)c;
    for (size_t i = lo; i < hi; ++i) {
        if (x->kind == 0) pedersen_eval(x->p, x->msgs + i * x->msg_len, x->msg_len, x->out + 2 * i);
        else bh_eval(x->p, x->msgs + i * x->msg_len, x->msg_len, x->out + i);
    }
}
/* returns 1 (bad length; the reference panics, pedersen/mod.rs:82-89) */
int orc_pedersen_crh_batch(void *h, const uint8_t *msgs, size_t n, size_t msg_len, uint64_t *out_xy, int threads) {
    const orc_curve_params *p = (const orc_curve_params *)h;
    if (msg_len * 8 > (size_t)p->window_size * p->num_windows) return 1;
    cb_ctx c = {p, msgs, msg_len, (fr *)out_xy, 0};
    parallel_for(n, threads, curve_range, &c);
    return 0;
}
int orc_bh_crh_batch(void *h, const uint8_t *msgs, size_t n, size_t msg_len, uint64_t *out_x, int threads) {
    const orc_curve_params *p = (const orc_curve_params *)h;
    if (msg_len * 8 > (size_t)p->window_size * p->num_windows * 3) return 1; /* bowe_hopwood/mod.rs:121-129 */
    cb_ctx c = {p, msgs, msg_len, (fr *)out_x, 1};
    parallel_for(n, threads, curve_range, &c);
    return 0;
}

/* byte-leaf Merkle tree with ByteDigestConverter (merkle_tree/mod.rs:67-78, tests/mod.rs:24-33).
 * kind 0 = Pedersen (digest = affine point, 64 B uncompressed x||y canonical LE),
 * kind 1 = Bowe-Hopwood (digest = Fq x, 32 B canonical LE).
 * two-to-one evaluate: zero buffer of (W*N)/8 bytes, left||right zip-truncated
 * (pedersen/mod.rs:158-182, bowe_hopwood/mod.rs:202-227). */
typedef struct {
    const orc_curve_params *p; int kind; const fr *child; fr *nodes; size_t first;
} clvl_ctx;
static void digest_bytes(int kind, const fr *d, uint8_t *out) {
    fr c;
    int nfe = kind == 0 ? 2 : 1;
    for (int k = 0; k < nfe; ++k) { fr_from_mont(&c, &d[k]); memcpy(out + 32 * k, c.l, 32); }
}
static void curve_level_range(void *cc, size_t lo, size_t hi) {
    clvl_ctx *x = (clvl_ctx *)cc;
    int nfe = x->kind == 0 ? 2 : 1;
    size_t dlen = 32 * (size_t)nfe;
    size_t buflen = ((size_t)x->p->window_size * x->p->num_windows) / 8;
    uint8_t *buf = (uint8_t *)malloc(buflen + 2 * dlen);
    uint8_t *src = (uint8_t *)malloc(2 * dlen);
    for (size_t i = lo; i < hi; ++i) {
        digest_bytes(x->kind, &x->child[(2 * i) * nfe], src);
        digest_bytes(x->kind, &x->child[(2 * i + 1) * nfe], src + dlen);
        memset(buf, 0, buflen);
        memcpy(buf, src, 2 * dlen < buflen ? 2 * dlen : buflen);
        if (x->kind == 0) pedersen_eval(x->p, buf, buflen, &x->nodes[(x->first + i) * 2]);
        else bh_eval(x->p, buf, buflen, &x->nodes[x->first + i]);
    }
    free(buf); free(src);
}
int orc_curve_merkle_build(int kind, void *leaf_h, void *two_h, const uint8_t *leaves, size_t n_leaves, size_t leaf_len,
                           uint64_t *leaf_nodes, uint64_t *non_leaf_nodes, int threads) {
    if (n_leaves < 2 || (n_leaves & (n_leaves - 1))) return 1;
    int rc = kind == 0 ? orc_pedersen_crh_batch(leaf_h, leaves, n_leaves, leaf_len, leaf_nodes, threads)
                       : orc_bh_crh_batch(leaf_h, leaves, n_leaves, leaf_len, leaf_nodes, threads);
    if (rc) return rc;
    int nfe = kind == 0 ? 2 : 1;
    fr *nl = (fr *)non_leaf_nodes;
    size_t width = n_leaves / 2, first = width - 1;
    clvl_ctx c = {(const orc_curve_params *)two_h, kind, (const fr *)leaf_nodes, nl, first};
    parallel_for(width, threads, curve_level_range, &c);
    while (width > 1) {
        size_t child_first = first;
        width /= 2; first = width - 1;
        clvl_ctx u = {(const orc_curve_params *)two_h, kind, nl + child_first * nfe, nl, first};
        parallel_for(width, threads, curve_level_range, &u);
    }
    return 0;
}

/* point helpers exported for tests */
void orc_te_add_affine(const uint64_t *p_xy, const uint64_t *q_xy, uint64_t *out_xy) {
    te_init();
    tep a, b, r;
    te_from_affine(&a, (const fr *)p_xy, (const fr *)(p_xy + 4));
    te_from_affine(&b, (const fr *)q_xy, (const fr *)(q_xy + 4));
    te_add(&r, &a, &b);
    te_to_affine(&r, (fr *)out_xy, (fr *)(out_xy + 4));
}
int orc_hardware_threads(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}
