/*
 * oracle/pasta_ref.c -- CPU restatement (plain C, 4 x u64 Montgomery limbs,
 * unsigned __int128) of the reference's MSM + NTT hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The product library (libkimchi_hip.so)
 * never links or calls anything in here.
 *
 * What it restates (results are unique mathematical objects; the arithmetic
 * crates ark-ff/ark-ec/ark-poly 0.5.0 are un-vendored, Cargo.lock:171-281):
 *   - Fp/Fq Montgomery arithmetic ....... curves/src/pasta/fields/fp.rs:8-80, fq.rs:8-79
 *   - Pallas/Vesta group law ............ curves/src/pasta/curves/{pallas,vesta}.rs
 *   - VariableBaseMSM::msm_bigint ....... called at poly-commitment/src/ipa.rs:649-672,
 *                                         commitment.rs:382 (signed-window Pippenger; threads over
 *                                         (window, point slice) jobs: ark-ec's rayon path over the
 *                                         windows + the point split of ipa.rs:652-662)
 *   - Radix2EvaluationDomain fft/ifft ... called at kimchi/src/prover.rs:289,377,907,
 *                                         circuits/constraints.rs:490-495 (threads over the columns
 *                                         of a batch, or over the butterflies of a stage when there
 *                                         are fewer columns than threads)
 *   - SRS::create ........................ poly-commitment/src/ipa.rs:751-778, 234-265,
 *                                         groupmap/src/lib.rs:74-189
 *   - compressed point codec ............ utils/src/serialization.rs:67-104
 * Pinned by tests/test_oracle_kats.py against tests/golden/reference_kats.json
 * (vectors parsed out of the reference by tests/golden/make_golden.py) and
 * cross-checked against the independent Python big-int oracle (oracle/pasta.py).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 l[4]; } fe;

typedef struct {
    fe p;        /* modulus */
    u64 inv;     /* -p^-1 mod 2^64 */
    fe one;      /* R mod p */
    fe r2;       /* R^2 mod p */
    fe root;     /* 2-adic root of unity 5^T, Montgomery form */
    fe t_m1_d2;  /* (T-1)/2, plain integer */
    fe pm1_d2;   /* (p-1)/2, plain integer */
    fe five;     /* 5 in Montgomery form (curve b) */
} field_t;

static field_t F[2];          /* 0 = Fp, 1 = Fq */
static int g_init_done = 0;

/* ---------------------------------------------------------------- bigint */
static inline int ge4(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline int is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int eq4(const fe *a, const fe *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline u64 sub4(fe *r, const fe *a, const fe *b) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - (u64)br;
        r->l[i] = (u64)d;
        br = (d >> 64) & 1;
    }
    return (u64)br;
}
static inline u64 add4(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        r->l[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}

/* ---------------------------------------------------------------- field */
static inline void f_add(const field_t *f, fe *r, const fe *a, const fe *b) {
    fe t; add4(&t, a, b);               /* p < 2^255: no carry out */
    if (ge4(&t, &f->p)) sub4(&t, &t, &f->p);
    *r = t;
}
static inline void f_sub(const field_t *f, fe *r, const fe *a, const fe *b) {
    fe t;
    if (sub4(&t, a, b)) add4(&t, &t, &f->p);
    *r = t;
}
static inline void f_neg(const field_t *f, fe *r, const fe *a) {
    if (is_zero(a)) { *r = *a; return; }
    sub4(r, &f->p, a);
}
static inline void f_dbl(const field_t *f, fe *r, const fe *a) { f_add(f, r, a, a); }

/* CIOS Montgomery multiplication, R = 2^256; unrolled.  p < 2^255 and a, b < p keep the running value below 2p, so the fifth word is 0 or 1.
 * (The variant that folds the two carry words into one chain -- possible because the top bit of p is clear -- measured SLOWER with gcc and clang:
 * it serialises the product and the reduction of a round.) */
static inline void f_mul(const field_t *f, fe *r, const fe *a, const fe *b) {
    const u64 *A = a->l, *B = b->l, *P = f->p.l;
    const u64 inv = f->inv;
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define KO_MUL_ROUND(i)                                                        \
    do {                                                                       \
        u128 c = (u128)A[0] * B[i] + t0; t0 = (u64)c; c >>= 64;                \
        c += (u128)A[1] * B[i] + t1; t1 = (u64)c; c >>= 64;                    \
        c += (u128)A[2] * B[i] + t2; t2 = (u64)c; c >>= 64;                    \
        c += (u128)A[3] * B[i] + t3; t3 = (u64)c; c >>= 64;                    \
        c += t4; t4 = (u64)c; const u64 t5 = (u64)(c >> 64);                   \
        const u64 m = t0 * inv;                                                \
        c = (u128)m * P[0] + t0; c >>= 64;                                     \
        c += (u128)m * P[1] + t1; t0 = (u64)c; c >>= 64;                       \
        c += (u128)m * P[2] + t2; t1 = (u64)c; c >>= 64;                       \
        c += (u128)m * P[3] + t3; t2 = (u64)c; c >>= 64;                       \
        c += t4; t3 = (u64)c; t4 = t5 + (u64)(c >> 64);                        \
    } while (0)
    KO_MUL_ROUND(0); KO_MUL_ROUND(1); KO_MUL_ROUND(2); KO_MUL_ROUND(3);
#undef KO_MUL_ROUND
    fe out = {{t0, t1, t2, t3}};
    if (t4 || ge4(&out, &f->p)) sub4(&out, &out, &f->p);
    *r = out;
}
static inline void f_sqr(const field_t *f, fe *r, const fe *a) { f_mul(f, r, a, a); }

static void f_pow(const field_t *f, fe *r, const fe *a, const fe *e /* plain integer */) {
    fe acc = f->one, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) f_mul(f, &acc, &acc, &base);
        f_sqr(f, &base, &base);
    }
    *r = acc;
}
static void f_inv(const field_t *f, fe *r, const fe *a) {
    fe e = f->p; e.l[0] -= 2;           /* p-2: low limb ends ...0001 so no borrow */
    f_pow(f, r, a, &e);
}
static void f_to_mont(const field_t *f, fe *r, const fe *a) { f_mul(f, r, a, &f->r2); }
static void f_from_mont(const field_t *f, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    f_mul(f, r, a, &one);
}
static int f_is_square(const field_t *f, const fe *a) {
    if (is_zero(a)) return 1;
    fe r; f_pow(f, &r, a, &f->pm1_d2);
    return eq4(&r, &f->one);
}
/* Tonelli-Shanks as in ark-ff: no sign normalisation. Returns 0 if non-residue. */
static int f_sqrt(const field_t *f, fe *out, const fe *a) {
    if (is_zero(a)) { *out = *a; return 1; }
    if (!f_is_square(f, a)) return 0;
    fe z = f->root, w, x, b;
    f_pow(f, &w, a, &f->t_m1_d2);
    f_mul(f, &x, a, &w);
    f_mul(f, &b, &x, &w);
    int v = 32;
    while (!eq4(&b, &f->one)) {
        int k = 0; fe b2k = b;
        while (!eq4(&b2k, &f->one)) { f_sqr(f, &b2k, &b2k); k++; }
        w = z;
        for (int i = 0; i < v - k - 1; i++) f_sqr(f, &w, &w);
        f_sqr(f, &z, &w);
        f_mul(f, &b, &b, &z);
        f_mul(f, &x, &x, &w);
        v = k;
    }
    *out = x;
    return 1;
}

/* ---------------------------------------------------------------- init */
static void set_hex(fe *r, const char *hex) {   /* 64 hex digits, big-endian */
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int j = 0; j < 16; j++) {
            char c = hex[(3 - i) * 16 + j];
            v = (v << 4) | (u64)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
        }
        r->l[i] = v;
    }
}
static void field_init(field_t *f, const char *phex) {
    set_hex(&f->p, phex);
    u64 inv = 1;                          /* Newton: p * inv == 1 mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - f->p.l[0] * inv;
    f->inv = (u64)0 - inv;
    /* R mod p by 256 modular doublings of 1; R2 by 256 more */
    fe r = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; i++) f_add(f, &r, &r, &r);
    f->one = r;
    for (int i = 0; i < 256; i++) f_add(f, &r, &r, &r);
    f->r2 = r;
    /* (p-1)/2, T=(p-1)>>32, (T-1)/2 */
    fe pm1 = f->p; pm1.l[0] -= 1;
    for (int i = 0; i < 4; i++) f->pm1_d2.l[i] = (pm1.l[i] >> 1) | (i < 3 ? pm1.l[i + 1] << 63 : 0);
    fe T;
    for (int i = 0; i < 4; i++) T.l[i] = (pm1.l[i] >> 32) | (i < 3 ? pm1.l[i + 1] << 32 : 0);
    fe Tm1 = T; Tm1.l[0] -= 1;            /* T is odd */
    for (int i = 0; i < 4; i++) f->t_m1_d2.l[i] = (Tm1.l[i] >> 1) | (i < 3 ? Tm1.l[i + 1] << 63 : 0);
    fe five = {{5, 0, 0, 0}};
    f_to_mont(f, &f->five, &five);
    f_pow(f, &f->root, &f->five, &T);     /* generator 5 (fp.rs:9, fq.rs:9) */
}
void ko_init(void) {
    if (g_init_done) return;
    field_init(&F[0], "40000000000000000000000000000000224698fc094cf91b992d30ed00000001");
    field_init(&F[1], "40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001");
    g_init_done = 1;
}
/* curve 0 = Vesta (coords Fq, scalars Fp); curve 1 = Pallas (coords Fp, scalars Fq) */
static inline const field_t *base_field(int curve) { ko_init(); return &F[curve == 0 ? 1 : 0]; }
static inline const field_t *scalar_field(int curve) { ko_init(); return &F[curve == 0 ? 0 : 1]; }

/* ---------------------------------------------------------------- group (Jacobian) */
typedef struct { fe x, y, z; } jac;       /* z == 0 <=> infinity */
typedef struct { fe x, y; } aff;

static inline void j_set_inf(const field_t *f, jac *p) { p->x = f->one; p->y = f->one; memset(&p->z, 0, sizeof(fe)); }
static inline int j_is_inf(const jac *p) { return is_zero(&p->z); }

static void j_dbl(const field_t *f, jac *r, const jac *p) {
    if (j_is_inf(p) || is_zero(&p->y)) { j_set_inf(f, r); return; }
    fe A, B, C, D, E, Fq_, t, X3, Y3, Z3;
    f_sqr(f, &A, &p->x); f_sqr(f, &B, &p->y); f_sqr(f, &C, &B);
    f_add(f, &t, &p->x, &B); f_sqr(f, &t, &t); f_sub(f, &t, &t, &A); f_sub(f, &t, &t, &C);
    f_dbl(f, &D, &t);
    f_dbl(f, &E, &A); f_add(f, &E, &E, &A);
    f_sqr(f, &Fq_, &E);
    f_sub(f, &X3, &Fq_, &D); f_sub(f, &X3, &X3, &D);
    f_sub(f, &t, &D, &X3); f_mul(f, &Y3, &E, &t);
    f_dbl(f, &C, &C); f_dbl(f, &C, &C); f_dbl(f, &C, &C);
    f_sub(f, &Y3, &Y3, &C);
    f_mul(f, &Z3, &p->y, &p->z); f_dbl(f, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void j_add(const field_t *f, jac *r, const jac *p, const jac *q) {
    if (j_is_inf(p)) { *r = *q; return; }
    if (j_is_inf(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t, X3, Y3, Z3;
    f_sqr(f, &z1z1, &p->z); f_sqr(f, &z2z2, &q->z);
    f_mul(f, &u1, &p->x, &z2z2); f_mul(f, &u2, &q->x, &z1z1);
    f_mul(f, &s1, &p->y, &q->z); f_mul(f, &s1, &s1, &z2z2);
    f_mul(f, &s2, &q->y, &p->z); f_mul(f, &s2, &s2, &z1z1);
    if (eq4(&u1, &u2)) {
        if (eq4(&s1, &s2)) { j_dbl(f, r, p); return; }
        j_set_inf(f, r); return;
    }
    f_sub(f, &h, &u2, &u1); f_sub(f, &rr, &s2, &s1);
    f_sqr(f, &hh, &h); f_mul(f, &hhh, &h, &hh); f_mul(f, &v, &u1, &hh);
    f_sqr(f, &X3, &rr); f_sub(f, &X3, &X3, &hhh); f_sub(f, &X3, &X3, &v); f_sub(f, &X3, &X3, &v);
    f_sub(f, &t, &v, &X3); f_mul(f, &Y3, &rr, &t); f_mul(f, &t, &s1, &hhh); f_sub(f, &Y3, &Y3, &t);
    f_mul(f, &Z3, &p->z, &q->z); f_mul(f, &Z3, &Z3, &h);
    r->x = X3; r->y = Y3; r->z = Z3;
}
/* mixed add: q affine, never infinity; neg != 0 adds -q */
static void j_madd(const field_t *f, jac *r, const jac *p, const aff *q, int neg) {
    fe qy = q->y;
    if (neg) f_neg(f, &qy, &qy);
    if (j_is_inf(p)) { r->x = q->x; r->y = qy; r->z = f->one; return; }
    fe z1z1, u2, s2, h, rr, hh, hhh, v, t, X3, Y3, Z3;
    f_sqr(f, &z1z1, &p->z);
    f_mul(f, &u2, &q->x, &z1z1);
    f_mul(f, &s2, &qy, &p->z); f_mul(f, &s2, &s2, &z1z1);
    if (eq4(&p->x, &u2)) {
        if (eq4(&p->y, &s2)) { j_dbl(f, r, p); return; }
        j_set_inf(f, r); return;
    }
    f_sub(f, &h, &u2, &p->x); f_sub(f, &rr, &s2, &p->y);
    f_sqr(f, &hh, &h); f_mul(f, &hhh, &h, &hh); f_mul(f, &v, &p->x, &hh);
    f_sqr(f, &X3, &rr); f_sub(f, &X3, &X3, &hhh); f_sub(f, &X3, &X3, &v); f_sub(f, &X3, &X3, &v);
    f_sub(f, &t, &v, &X3); f_mul(f, &Y3, &rr, &t); f_mul(f, &t, &p->y, &hhh); f_sub(f, &Y3, &Y3, &t);
    f_mul(f, &Z3, &p->z, &h);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static int j_to_affine(const field_t *f, aff *r, const jac *p) {  /* returns 1 if infinity */
    if (j_is_inf(p)) { memset(r, 0, sizeof(*r)); return 1; }
    fe zi, zi2, zi3;
    f_inv(f, &zi, &p->z); f_sqr(f, &zi2, &zi); f_mul(f, &zi3, &zi2, &zi);
    f_mul(f, &r->x, &p->x, &zi2); f_mul(f, &r->y, &p->y, &zi3);
    return 0;
}
static void j_mul_plain(const field_t *f, jac *r, const jac *p, const fe *k /* plain integer */) {
    jac acc; j_set_inf(f, &acc);
    for (int i = 255; i >= 0; i--) {
        j_dbl(f, &acc, &acc);
        if ((k->l[i >> 6] >> (i & 63)) & 1) j_add(f, &acc, &acc, p);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- exported: field batch ops */
int ko_field_op(int field, int op, const u64 *a, const u64 *b, u64 *out, size_t n) {
    ko_init();
    const field_t *f = &F[field & 1];
    for (size_t i = 0; i < n; i++) {
        fe x, y, r;
        memcpy(&x, a + 4 * i, 32);
        if (b) memcpy(&y, b + 4 * i, 32); else memset(&y, 0, 32);
        switch (op) {
            case 0: f_mul(f, &r, &x, &y); break;
            case 1: f_add(f, &r, &x, &y); break;
            case 2: f_sub(f, &r, &x, &y); break;
            case 3: f_to_mont(f, &r, &x); break;
            case 4: f_from_mont(f, &r, &x); break;
            case 5: f_inv(f, &r, &x); break;
            case 6: f_sqr(f, &r, &x); break;
            default: return -1;
        }
        memcpy(out + 4 * i, &r, 32);
    }
    return 0;
}
void ko_field_consts(int field, u64 *p, u64 *one, u64 *r2, u64 *inv, u64 *root) {
    ko_init();
    const field_t *f = &F[field & 1];
    memcpy(p, &f->p, 32); memcpy(one, &f->one, 32); memcpy(r2, &f->r2, 32);
    *inv = f->inv; memcpy(root, &f->root, 32);
}

/* ---------------------------------------------------------------- exported: point ops (affine in/out, Montgomery coords) */
static void load_aff(jac *j, const field_t *f, const u64 *xy, int inf) {
    if (inf) { j_set_inf(f, j); return; }
    memcpy(&j->x, xy, 32); memcpy(&j->y, xy + 4, 32); j->z = f->one;
}
static void store_aff(const field_t *f, const jac *j, u64 *xy, uint8_t *inf) {
    aff a; int i = j_to_affine(f, &a, j);
    memcpy(xy, &a.x, 32); memcpy(xy + 4, &a.y, 32);
    if (inf) *inf = (uint8_t)i;
}
int ko_point_add(int curve, const u64 *p_xy, int p_inf, const u64 *q_xy, int q_inf, u64 *out_xy, uint8_t *out_inf) {
    const field_t *f = base_field(curve);
    jac p, q, r; load_aff(&p, f, p_xy, p_inf); load_aff(&q, f, q_xy, q_inf);
    j_add(f, &r, &p, &q); store_aff(f, &r, out_xy, out_inf);
    return 0;
}
/* scalar: 4 limbs; scalar_is_mont selects Montgomery vs canonical */
int ko_point_mul(int curve, const u64 *p_xy, int p_inf, const u64 *scalar, int scalar_is_mont, u64 *out_xy, uint8_t *out_inf) {
    const field_t *f = base_field(curve), *sf = scalar_field(curve);
    jac p, r; load_aff(&p, f, p_xy, p_inf);
    fe k; memcpy(&k, scalar, 32);
    if (scalar_is_mont) f_from_mont(sf, &k, &k);
    j_mul_plain(f, &r, &p, &k); store_aff(f, &r, out_xy, out_inf);
    return 0;
}
int ko_is_on_curve(int curve, const u64 *xy) {
    const field_t *f = base_field(curve);
    fe x, y, l, r; memcpy(&x, xy, 32); memcpy(&y, xy + 4, 32);
    f_sqr(f, &l, &y); f_sqr(f, &r, &x); f_mul(f, &r, &r, &x); f_add(f, &r, &r, &f->five);
    return eq4(&l, &r);
}

/* ---------------------------------------------------------------- MSM */
/* Signed-window Pippenger (what VariableBaseMSM::msm does in ark-ec, commitment.rs:382), scheduled for a many-core host:
 *   - digits precomputed in parallel (n * nwin int32, signed digits in [-2^(c-1), 2^(c-1)]);
 *   - job = (window w, point slice s): its own bucket set over points [n*s/slices, n*(s+1)/slices); the window width is chosen so that
 *     a bucket set stays cache-resident and there are at least as many jobs as threads (the reference parallelises over windows with rayon
 *     and splits the points in two, ipa.rs:652-662; with 2 x 128 hardware threads that alone would leave most of the machine idle);
 *   - jobs are handed out from a shared counter (a slice with many zero digits finishes early). */
typedef struct {
    const field_t *f, *sf; const aff *pts; const uint8_t *inf; const u64 *scalars; int32_t *digits; size_t n;
    int c, nwin, slices, njobs, mont, nthreads; jac *win_sums;   /* win_sums[w * slices + s] */
    volatile int next_job;
} msm_shared;
typedef struct { msm_shared *S; int tid; } msm_arg;

static void msm_digits_range(msm_shared *S, size_t i0, size_t i1) {
    const int c = S->c, nwin = S->nwin; const size_t nb = (size_t)1 << (c - 1), n = S->n;
    for (size_t i = i0; i < i1; i++) {
        fe s; memcpy(&s, S->scalars + 4 * i, 32);
        if (S->mont) f_from_mont(S->sf, &s, &s);
        else if (ge4(&s, &S->sf->p)) { /* reduce non-canonical input once */ sub4(&s, &s, &S->sf->p); }
        int carry = 0;
        const int skip = S->inf && S->inf[i];
        for (int w = 0; w < nwin; w++) {
            int bit = w * c; u64 raw = 0;
            if (bit < 256) {
                raw = s.l[bit >> 6] >> (bit & 63);
                if ((bit & 63) + c > 64 && (bit >> 6) < 3) raw |= s.l[(bit >> 6) + 1] << (64 - (bit & 63));
                raw &= ((u64)1 << c) - 1;
            }
            int64_t v = (int64_t)raw + carry;
            if (v > (int64_t)nb) { v -= (int64_t)1 << c; carry = 1; } else carry = 0;
            S->digits[(size_t)w * n + i] = skip ? 0 : (int32_t)v;
        }
    }
}
static void *msm_digit_worker(void *arg) {
    msm_arg *A = (msm_arg *)arg; msm_shared *S = A->S;
    msm_digits_range(S, S->n * (size_t)A->tid / (size_t)S->nthreads, S->n * (size_t)(A->tid + 1) / (size_t)S->nthreads);
    return NULL;
}
static void *msm_bucket_worker(void *arg) {
    msm_shared *S = ((msm_arg *)arg)->S;
    const field_t *f = S->f;
    const size_t nb = (size_t)1 << (S->c - 1);
    jac *buckets = (jac *)malloc(sizeof(jac) * nb);
    for (;;) {
        const int job = __atomic_fetch_add(&S->next_job, 1, __ATOMIC_RELAXED);
        if (job >= S->njobs) break;
        const int w = job / S->slices, sl = job % S->slices;
        const size_t i0 = S->n * (size_t)sl / (size_t)S->slices, i1 = S->n * (size_t)(sl + 1) / (size_t)S->slices;
        for (size_t b = 0; b < nb; b++) j_set_inf(f, &buckets[b]);
        const int32_t *dg = S->digits + (size_t)w * S->n;
        for (size_t i = i0; i < i1; i++) {
            const int32_t d = dg[i];
            if (d == 0) continue;
            const size_t b = (size_t)(d < 0 ? -d : d) - 1;
            j_madd(f, &buckets[b], &buckets[b], &S->pts[i], d < 0);
        }
        jac run, acc; j_set_inf(f, &run); j_set_inf(f, &acc);
        for (size_t b = nb; b-- > 0;) {
            j_add(f, &run, &run, &buckets[b]);
            j_add(f, &acc, &acc, &run);
        }
        S->win_sums[job] = acc;
    }
    free(buckets);
    return NULL;
}

static int g_last_threads = 1;
int ko_last_threads(void) { return g_last_threads; }

/* window width and point slices for n points on `threads` threads: the cheapest schedule among c = 4..16 and the slice counts around
 * threads / windows (up to ~4 jobs per thread) -- per thread, in field multiplications: the mixed additions of the jobs it runs (11 products + the additions and
 * subtractions around them ~ 14) + the running sums of their bucket sets (two full additions of 16 products per bucket) */
static void pick_schedule(size_t n, int threads, int *c_out, int *slices_out) {
    double best = 0; int bc = 3, bs = 1;
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
    if (lg < 6) { *c_out = 3; *slices_out = 1; return; }
    for (int c = 4; c <= 16; c++) {
        const int nwin = (255 + c - 1) / c + 1;
        const double nb = (double)((size_t)1 << (c - 1));
        const int smax = 4 * threads / nwin + 2;                          /* up to ~4 jobs per thread: the queue evens out what remains */
        for (int slices = 1; slices <= smax; slices++) {
            if (slices > 1 && (double)n / slices < 4 * nb) break;           /* a slice should fill its buckets a few times over */
            const int njobs = nwin * slices;
            const int rounds = (njobs + threads - 1) / threads;              /* jobs one thread runs, worst case */
            const double cost = rounds * ((double)n / slices * 14.0 + nb * 32.0);
            if (best == 0 || cost < best) { best = cost; bc = c; bs = slices; }
        }
    }
    *c_out = bc; *slices_out = bs;
}

/*
 * Sum_i scalars[i] * P_i.  xy: n x 8 limbs (x||y, Montgomery, base field);
 * inf: nullable per-point infinity flags; scalars: n x 4 limbs.
 * Output affine (Montgomery) + infinity flag.  threads <= 0 -> 1.
 */
int ko_msm(int curve, const u64 *xy, const uint8_t *inf, const u64 *scalars, size_t n,
           int scalars_are_montgomery, int threads, u64 *out_xy, uint8_t *out_inf) {
    const field_t *f = base_field(curve), *sf = scalar_field(curve);
    jac total; j_set_inf(f, &total);
    if (n == 0) { store_aff(f, &total, out_xy, out_inf); return 0; }
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    {   /* starting a thread costs tens of microseconds: no more threads than there are ~16K-addition (a few ms) shares of the work */
        const size_t shares = n * 22 / 16384 + 1;
        if ((size_t)threads > shares) threads = (int)shares;
    }
    int c, slices; pick_schedule(n, threads, &c, &slices);
    const int nwin = (255 + c - 1) / c + 1;      /* +1: room for the final carry */
    const int njobs = nwin * slices;
    msm_shared S = {f, sf, (const aff *)xy, inf, scalars, NULL, n, c, nwin, slices, njobs, scalars_are_montgomery, threads, NULL, 0};
    S.digits = (int32_t *)malloc(sizeof(int32_t) * n * (size_t)nwin);
    S.win_sums = (jac *)malloc(sizeof(jac) * (size_t)njobs);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    msm_arg *args = (msm_arg *)malloc(sizeof(msm_arg) * (size_t)threads);
    for (int t = 0; t < threads; t++) args[t] = (msm_arg){&S, t};
    const int dthreads = n < 4096 ? 1 : ((size_t)threads > n / 4096 ? (int)(n / 4096) : threads);
    S.nthreads = dthreads;
    if (dthreads == 1) msm_digit_worker(&args[0]);
    else {
        for (int t = 0; t < dthreads; t++) pthread_create(&th[t], NULL, msm_digit_worker, &args[t]);
        for (int t = 0; t < dthreads; t++) pthread_join(th[t], NULL);
    }
    const int bthreads = threads < njobs ? threads : njobs;
    if (bthreads == 1) msm_bucket_worker(&args[0]);
    else {
        for (int t = 0; t < bthreads; t++) pthread_create(&th[t], NULL, msm_bucket_worker, &args[t]);
        for (int t = 0; t < bthreads; t++) pthread_join(th[t], NULL);
    }
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) j_dbl(f, &total, &total);
        for (int sl = 0; sl < slices; sl++) j_add(f, &total, &total, &S.win_sums[w * slices + sl]);
    }
    g_last_threads = bthreads;
    store_aff(f, &total, out_xy, out_inf);
    free(args); free(th); free(S.win_sums); free(S.digits);
    return 0;
}

/* Definition-level MSM: double-and-add per term. O(256 n) group ops. */
int ko_msm_naive(int curve, const u64 *xy, const uint8_t *inf, const u64 *scalars, size_t n,
                 int scalars_are_montgomery, u64 *out_xy, uint8_t *out_inf) {
    const field_t *f = base_field(curve), *sf = scalar_field(curve);
    jac total; j_set_inf(f, &total);
    for (size_t i = 0; i < n; i++) {
        jac p, r; load_aff(&p, f, xy + 8 * i, inf ? inf[i] : 0);
        fe k; memcpy(&k, scalars + 4 * i, 32);
        if (scalars_are_montgomery) f_from_mont(sf, &k, &k);
        j_mul_plain(f, &r, &p, &k);
        j_add(f, &total, &total, &r);
    }
    store_aff(f, &total, out_xy, out_inf);
    return 0;
}

/* ---------------------------------------------------------------- NTT */
static void root_of_unity(const field_t *f, fe *w, unsigned log2_n, int inverse) {
    *w = f->root;
    for (unsigned i = log2_n; i < 32; i++) f_sqr(f, w, w);
    if (inverse) f_inv(f, w, w);
}
static inline size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static void ntt_one(const field_t *f, fe *a, unsigned log2_n, const fe *tw /* stage-major twiddles, see ko_ntt */, int inverse, const fe *ninv) {
    size_t n = (size_t)1 << log2_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, log2_n);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (size_t m = 1; m < n; m <<= 1) {
        const fe *twm = tw + m;                              /* stage-major table: twm[j] = w^(j n / 2m), contiguous */
        for (size_t k = 0; k < n; k += 2 * m)
            for (size_t j = 0; j < m; j++) {
                fe u = a[k + j], v;
                f_mul(f, &v, &a[k + j + m], &twm[j]);
                f_add(f, &a[k + j], &u, &v);
                f_sub(f, &a[k + j + m], &u, &v);
            }
    }
    if (inverse) for (size_t i = 0; i < n; i++) f_mul(f, &a[i], &a[i], ninv);
}
typedef struct { const field_t *f; fe *data; unsigned log2_n; const fe *tw; int inverse; const fe *ninv; size_t b0, b1; } ntt_job;
static void *ntt_worker(void *arg) {
    ntt_job *J = (ntt_job *)arg;
    size_t n = (size_t)1 << J->log2_n;
    for (size_t b = J->b0; b < J->b1; b++) ntt_one(J->f, J->data + b * n, J->log2_n, J->tw, J->inverse, J->ninv);
    return NULL;
}
/* more threads than transforms: all threads walk all transforms stage by stage (the butterflies of a stage are independent), a barrier between
 * stages -- the parallel FFT of ark-poly splits a large transform over the rayon pool too */
typedef struct {
    const field_t *f; fe *data; unsigned log2_n; const fe *tw; int inverse; const fe *ninv; size_t batch; int tid, nthreads; pthread_barrier_t *bar;
} ntt_flat_job;
static void *ntt_flat_worker(void *arg) {
    ntt_flat_job *J = (ntt_flat_job *)arg;
    const field_t *f = J->f;
    const size_t n = (size_t)1 << J->log2_n, half = n / 2, T = (size_t)J->nthreads, t = (size_t)J->tid;
    const size_t all = J->batch * n;
    for (size_t x = all * t / T; x < all * (t + 1) / T; x++) {
        const size_t b = x / n, i = x % n, j = bitrev(i, J->log2_n);
        if (i < j) { fe *a = J->data + b * n; fe tmp = a[i]; a[i] = a[j]; a[j] = tmp; }
    }
    pthread_barrier_wait(J->bar);
    const size_t total = J->batch * half;
    unsigned lm = 0;
    for (size_t m = 1; m < n; m <<= 1, lm++) {
        const fe *twm = J->tw + m;
        for (size_t x = total * t / T; x < total * (t + 1) / T; x++) {
            const size_t b = x >> (J->log2_n - 1), r = x & (half - 1), k = (r >> lm) << (lm + 1), j = r & (m - 1);
            fe *a = J->data + b * n;
            fe u = a[k + j], v;
            f_mul(f, &v, &a[k + j + m], &twm[j]);
            f_add(f, &a[k + j], &u, &v);
            f_sub(f, &a[k + j + m], &u, &v);
        }
        pthread_barrier_wait(J->bar);
    }
    if (J->inverse) for (size_t x = all * t / T; x < all * (t + 1) / T; x++) f_mul(f, &J->data[x], &J->data[x], J->ninv);
    return NULL;
}
/* data: batch x N x 4 limbs, Montgomery, natural order in and out; inverse includes 1/N. */
int ko_ntt(int field, u64 *data, unsigned log2_n, int inverse, size_t batch, int threads) {
    ko_init();
    if (log2_n > 32) return -1;
    const field_t *f = &F[field & 1];
    size_t n = (size_t)1 << log2_n;
    fe w; root_of_unity(f, &w, log2_n, inverse);
    size_t half = n > 1 ? n / 2 : 1;
    /* the n/2 powers of w, then regrouped stage by stage: tw[m + j] = w^(j n / 2m) for the stage of half-size m (a butterfly loop then walks
     * its twiddles contiguously instead of with a stride of n / 2m elements -- at 8 KB strides every access is a new line in one cache set) */
    fe *pw = (fe *)malloc(sizeof(fe) * half);
    pw[0] = f->one;
    for (size_t i = 1; i < half; i++) f_mul(f, &pw[i], &pw[i - 1], &w);
    fe *tw = (fe *)malloc(sizeof(fe) * (n > 1 ? n : 2));
    tw[0] = f->one;
    for (size_t m = 1; m < n; m <<= 1) { const size_t step = n / (2 * m); for (size_t j = 0; j < m; j++) tw[m + j] = pw[j * step]; }
    free(pw);
    fe nn = {{n, 0, 0, 0}}, ninv; f_to_mont(f, &nn, &nn); f_inv(f, &ninv, &nn);
    if (threads < 1) threads = 1;
    if ((size_t)threads > batch && log2_n >= 12) {          /* fewer transforms than threads, and large enough to be worth the barriers */
        if ((size_t)threads > batch * (n / 4096)) threads = (int)(batch * (n / 4096));      /* >= 2048 butterflies per thread and stage */
        pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)threads);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        ntt_flat_job *jobs = (ntt_flat_job *)malloc(sizeof(ntt_flat_job) * (size_t)threads);
        for (int t = 0; t < threads; t++) {
            jobs[t] = (ntt_flat_job){f, (fe *)data, log2_n, tw, inverse, &ninv, batch, t, threads, &bar};
            pthread_create(&th[t], NULL, ntt_flat_worker, &jobs[t]);
        }
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&bar);
        free(jobs); free(th); free(tw);
        return 0;
    }
    if ((size_t)threads > batch) threads = (int)batch;
    if (threads <= 1) {
        ntt_job J = {f, (fe *)data, log2_n, tw, inverse, &ninv, 0, batch};
        ntt_worker(&J);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        ntt_job *jobs = (ntt_job *)malloc(sizeof(ntt_job) * (size_t)threads);
        for (int t = 0; t < threads; t++) {
            jobs[t] = (ntt_job){f, (fe *)data, log2_n, tw, inverse, &ninv, batch * t / threads, batch * (t + 1) / threads};
            pthread_create(&th[t], NULL, ntt_worker, &jobs[t]);
        }
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        free(jobs); free(th);
    }
    free(tw);
    return 0;
}
/* evaluate_over_domain: zero-extend n coeffs to n<<b, forward NTT. out: batch x (n<<b) x 4 limbs. */
typedef struct { const u64 *coeffs; u64 *out; size_t n, N, b0, b1; } lde_pad_job;
static void *lde_pad_worker(void *arg) {
    lde_pad_job *J = (lde_pad_job *)arg;
    for (size_t b = J->b0; b < J->b1; b++) {
        memcpy(J->out + b * J->N * 4, J->coeffs + b * J->n * 4, J->n * 32);
        memset(J->out + b * J->N * 4 + J->n * 4, 0, (J->N - J->n) * 32);
    }
    return NULL;
}
int ko_lde(int field, const u64 *coeffs, unsigned log2_n, unsigned log2_blowup, u64 *out, size_t batch, int threads) {
    size_t n = (size_t)1 << log2_n, N = n << log2_blowup;
    int pt = threads < 1 ? 1 : threads;
    if ((size_t)pt > batch) pt = (int)batch;
    if (pt <= 1 || N < 65536) { lde_pad_job J = {coeffs, out, n, N, 0, batch}; lde_pad_worker(&J); }
    else {                                                  /* the zero extension is 7/8 of the output: one thread per column */
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)pt);
        lde_pad_job *jobs = (lde_pad_job *)malloc(sizeof(lde_pad_job) * (size_t)pt);
        for (int t = 0; t < pt; t++) {
            jobs[t] = (lde_pad_job){coeffs, out, n, N, batch * (size_t)t / (size_t)pt, batch * (size_t)(t + 1) / (size_t)pt};
            pthread_create(&th[t], NULL, lde_pad_worker, &jobs[t]);
        }
        for (int t = 0; t < pt; t++) pthread_join(th[t], NULL);
        free(jobs); free(th);
    }
    return ko_ntt(field, out, log2_n + log2_blowup, 0, batch, threads);
}

/* ---------------------------------------------------------------- BLAKE2b-512 (RFC 7693), unkeyed */
static const u64 B2_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
/* single-block message (len <= 128) */
static void blake2b512_short(const uint8_t *msg, size_t len, uint8_t out[64]) {
    u64 h[8], m[16], v[16];
    uint8_t block[128];
    memset(block, 0, 128); memcpy(block, msg, len);
    for (int i = 0; i < 8; i++) h[i] = B2_IV[i];
    h[0] ^= 0x01010000ULL ^ 64;
    for (int i = 0; i < 16; i++) { u64 w = 0; for (int j = 7; j >= 0; j--) w = (w << 8) | block[8 * i + j]; m[i] = w; }
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= (u64)len; v[14] = ~v[14];
#define B2G(a, b, c, d, x, y) \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = B2_SIGMA[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]); B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2G(2, 6, 10, 14, m[s[4]], m[s[5]]); B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]); B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2G(2, 7, 8, 13, m[s[12]], m[s[13]]); B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (8 * j));
}
void ko_blake2b512(const uint8_t *msg, size_t len, uint8_t out[64]) { blake2b512_short(msg, len, out); }

/* ---------------------------------------------------------------- SRS::create (ipa.rs:751-778) */
typedef struct { fe u, fu, c1, s, c2; } bw_params;
static bw_params BW[2]; static int bw_done[2] = {0, 0};
static void bw_setup(int curve) {       /* groupmap/src/lib.rs:134-165, u = first x with x^3+5 != 0 */
    if (bw_done[curve]) return;
    const field_t *f = base_field(curve);
    bw_params *P = &BW[curve];
    P->u = f->one;
    fe three_u2, t, two;
    f_add(f, &P->fu, &f->one, &f->five);               /* u^3 + 5 = 6 */
    f_add(f, &two, &f->one, &f->one);
    f_add(f, &three_u2, &two, &f->one);                /* 3u^2 = 3 */
    f_inv(f, &P->c2, &three_u2);
    f_neg(f, &t, &three_u2);
    f_sqrt(f, &P->s, &t);
    f_sub(f, &t, &P->s, &P->u);
    fe two_inv; f_inv(f, &two_inv, &two);
    f_mul(f, &P->c1, &t, &two_inv);
    bw_done[curve] = 1;
}
static void curve_rhs(const field_t *f, fe *r, const fe *x) { f_sqr(f, r, x); f_mul(f, r, r, x); f_add(f, r, r, &f->five); }
static void point_of_random_bytes(int curve, const uint8_t *rb, aff *out) {
    const field_t *f = base_field(curve);
    const bw_params *P = &BW[curve];
    /* 31 bytes -> 248 bits, LSB-first per byte, read big-endian (ipa.rs:234-259) */
    fe t; memset(&t, 0, sizeof(t));
    for (int i = 0; i < 31; i++)
        for (int j = 0; j < 8; j++) {
            int bitpos = 247 - (8 * i + j);
            if ((rb[i] >> j) & 1) t.l[bitpos >> 6] |= (u64)1 << (bitpos & 63);
        }
    f_to_mont(f, &t, &t);
    fe t2, ai, alpha, x[3], tmp, tpf;
    f_sqr(f, &t2, &t);
    f_add(f, &tpf, &t2, &P->fu);
    f_mul(f, &ai, &tpf, &t2);
    if (is_zero(&ai)) memset(&alpha, 0, sizeof(alpha)); else f_inv(f, &alpha, &ai);
    f_sqr(f, &tmp, &t2); f_mul(f, &tmp, &tmp, &alpha); f_mul(f, &tmp, &tmp, &P->s);
    f_sub(f, &x[0], &P->c1, &tmp);
    f_neg(f, &tmp, &P->u); f_sub(f, &x[1], &tmp, &x[0]);
    fe t2inv; f_mul(f, &t2inv, &alpha, &tpf);
    f_sqr(f, &tmp, &tpf); f_mul(f, &tmp, &tmp, &t2inv); f_mul(f, &tmp, &tmp, &P->c2);
    f_sub(f, &x[2], &P->u, &tmp);
    for (int k = 0; k < 3; k++) {
        fe rhs, y; curve_rhs(f, &rhs, &x[k]);
        if (f_sqrt(f, &y, &rhs)) { out->x = x[k]; out->y = y; return; }
    }
    abort();
}
typedef struct { int curve; size_t i0, i1, start; u64 *out; } srs_job;
static void *srs_worker(void *arg) {
    srs_job *J = (srs_job *)arg;
    for (size_t i = J->i0; i < J->i1; i++) {
        uint32_t idx = (uint32_t)(J->start + i);
        uint8_t msg[4] = {(uint8_t)(idx >> 24), (uint8_t)(idx >> 16), (uint8_t)(idx >> 8), (uint8_t)idx};
        uint8_t dig[64]; blake2b512_short(msg, 4, dig);
        aff p; point_of_random_bytes(J->curve, dig, &p);
        memcpy(J->out + 8 * i, &p, 64);
    }
    return NULL;
}
/* g_{start} .. g_{start+count-1}, affine Montgomery x||y */
int ko_srs_generate(int curve, size_t start, size_t count, u64 *xy_out, int threads) {
    bw_setup(curve);
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = count ? (int)count : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    srs_job *jobs = (srs_job *)malloc(sizeof(srs_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (srs_job){curve, count * t / threads, count * (t + 1) / threads, start, xy_out};
        if (threads == 1) srs_worker(&jobs[t]); else pthread_create(&th[t], NULL, srs_worker, &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(jobs); free(th);
    return 0;
}
int ko_srs_h(int curve, u64 *xy_out) {       /* ipa.rs:765-772 */
    bw_setup(curve);
    uint8_t msg[12] = {'s', 'r', 's', '_', 'm', 'i', 's', 'c', 0, 0, 0, 0};
    uint8_t dig[64]; blake2b512_short(msg, 12, dig);
    aff p; point_of_random_bytes(curve, dig, &p);
    memcpy(xy_out, &p, 64);
    return 0;
}
/* ark-serialize compressed SW: 32-byte LE canonical x, flag byte (0x80 = y > (p-1)/2, 0x40 = infinity) */
int ko_compress(int curve, const u64 *xy, const uint8_t *inf, size_t n, uint8_t *out33) {
    const field_t *f = base_field(curve);
    for (size_t i = 0; i < n; i++) {
        uint8_t *o = out33 + 33 * i;
        if (inf && inf[i]) { memset(o, 0, 33); o[32] = 0x40; continue; }
        fe x, y; memcpy(&x, xy + 8 * i, 32); memcpy(&y, xy + 8 * i + 4, 32);
        f_from_mont(f, &x, &x); f_from_mont(f, &y, &y);
        memcpy(o, &x, 32);
        int neg = !ge4(&f->pm1_d2, &y);      /* y > (p-1)/2 */
        o[32] = neg ? 0x80 : 0x00;
    }
    return 0;
}

/* ---------------------------------------------------------------- Lagrange basis (ipa.rs:1065-1172), group iNTT */
int ko_lagrange_basis(int curve, const u64 *g_xy, size_t n_g, unsigned log2_n, unsigned chunk,
                      u64 *out_xy, uint8_t *out_inf) {
    const field_t *f = base_field(curve), *sf = scalar_field(curve);
    size_t n = (size_t)1 << log2_n;
    size_t start = (size_t)chunk * n_g;
    if (start >= n) return -1;
    size_t num_terms = ((size_t)(chunk + 1) * n_g < n ? (size_t)(chunk + 1) * n_g : n) - start;
    jac *a = (jac *)malloc(sizeof(jac) * n);
    for (size_t i = 0; i < n; i++) j_set_inf(f, &a[i]);
    for (size_t j = 0; j < num_terms; j++) load_aff(&a[start + j], f, g_xy + 8 * j, 0);
    fe w; root_of_unity(sf, &w, log2_n, 1);
    size_t half = n > 1 ? n / 2 : 1;
    fe *tw = (fe *)malloc(sizeof(fe) * half);     /* plain-integer twiddles for scalar mul */
    fe cur = sf->one;
    for (size_t i = 0; i < half; i++) { f_from_mont(sf, &tw[i], &cur); f_mul(sf, &cur, &cur, &w); }
    for (size_t i = 0; i < n; i++) { size_t j = bitrev(i, log2_n); if (i < j) { jac t = a[i]; a[i] = a[j]; a[j] = t; } }
    for (size_t m = 1; m < n; m <<= 1) {
        size_t step = n / (2 * m);
        for (size_t k = 0; k < n; k += 2 * m)
            for (size_t j = 0; j < m; j++) {
                jac u = a[k + j], v, nv;
                if (j == 0) v = a[k + j + m]; else j_mul_plain(f, &v, &a[k + j + m], &tw[j * step]);
                j_add(f, &a[k + j], &u, &v);
                nv = v; f_neg(f, &nv.y, &nv.y);
                j_add(f, &a[k + j + m], &u, &nv);
            }
    }
    fe nn = {{n, 0, 0, 0}}, ninv; f_to_mont(sf, &nn, &nn); f_inv(sf, &ninv, &nn); f_from_mont(sf, &ninv, &ninv);
    for (size_t i = 0; i < n; i++) {
        jac r; j_mul_plain(f, &r, &a[i], &ninv);
        store_aff(f, &r, out_xy + 8 * i, &out_inf[i]);
    }
    free(tw); free(a);
    return 0;
}
