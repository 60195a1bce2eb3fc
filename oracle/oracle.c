/*
 * oracle.c — CPU restatement (plain C; the hot path on a single thread, the general-level key switch's independent rows on OpenMP threads) of the reference's conv hot path. See oracle.h.
 * TEST INFRASTRUCTURE: never linked into or called by the product library.
 *
 * Citations: `file.go:a-b` = /root/reference/file.go; `lattigo:` = the pinned dependency
 * github.com/dwkim606/test_lattigo@eb33b0555aaa (fork of Lattigo v2.2.0), which is NOT in /root/reference;
 * for it the symbol in /root/reference/test_run and the SURVEY.md section 8(a)-R row are given.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct {
    uint64_t q, qinv;        /* q^-1 mod 2^64 (lattigo ring.MRedParams) */
    uint64_t bred_hi, bred_lo; /* floor(2^128/q) (lattigo ring.BRedParams) */
    uint64_t r_mod_q;        /* 2^64 mod q = MForm(1) */
    uint64_t r2_mod_q;       /* 2^128 mod q */
    uint64_t *psi, *psi_inv; /* Montgomery form, index bitrev(j) holds psi^j (ring.genNTTParams) */
    uint64_t n_inv;          /* MForm(N^-1) */
} or_mod;

struct or_ctx {
    int logN, N, nq, np;
    or_mod *m; /* nq + np */
};

/* ---------- scalar arithmetic ---------- */
/* lattigo ring.MRed (test_run:ring.reconstructRNS @0x4e79b0 shows the inlined body) */
static inline uint64_t mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t h = (uint64_t)(((u128)(mlo * qinv) * q) >> 64);
    uint64_t r = mhi - h + q;
    if (r >= q) r -= q;
    return r;
}
/* lattigo ring.MRedConstant: same without the last conditional subtraction, result in [0,2q) */
static inline uint64_t mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t h = (uint64_t)(((u128)(mlo * qinv) * q) >> 64);
    return mhi - h + q;
}
/* lattigo ring.BRedAdd: x mod q for a 64-bit x using the high word of floor(2^128/q) */
static inline uint64_t bred_add(uint64_t x, uint64_t q, uint64_t u_hi) {
    uint64_t s = (uint64_t)(((u128)x * u_hi) >> 64);
    uint64_t r = x - s * q;
    if (r >= q) r -= q;
    return r;
}
static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1; b %= q;
    while (e) { if (e & 1) r = mulmod(r, b, q); b = mulmod(b, b, q); e >>= 1; }
    return r;
}
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t r = a + b; if (r >= q) r -= q; return r; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { uint64_t r = a + q - b; if (r >= q) r -= q; return r; }
/* lattigo ring.MForm: a * 2^64 mod q */
static inline uint64_t mform(uint64_t a, const or_mod *m) { return (uint64_t)((((u128)a) << 64) % m->q); }

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* lattigo ring.primitiveRoot (test_run @0x4f3de0): g starts at 2 and is incremented BEFORE the first test,
 * so the first candidate is 3; accept the first g with g^((q-1)/f) != 1 for every prime factor f of q-1. */
uint64_t or_primitive_root(uint64_t q) {
    uint64_t fac[64]; int nf = 0; uint64_t n = q - 1;
    for (uint64_t p = 2; p * p <= n; p += (p == 2 ? 1 : 2)) {
        if (n % p == 0) { fac[nf++] = p; while (n % p == 0) n /= p; }
    }
    if (n > 1) fac[nf++] = n;
    for (uint64_t g = 3;; g++) {
        int ok = 1;
        for (int i = 0; i < nf; i++) if (powmod(g, (q - 1) / fac[i], q) == 1) { ok = 0; break; }
        if (ok) return g;
    }
}

/* lattigo ring.(*Ring).genNTTParams: psi = g^((q-1)/2N); tables in Montgomery form with
 * NttPsi[bitrev(j)] = psi^j, NttPsiInv[bitrev(j)] = psi^-j; NttNInv = MForm(N^-1). */
static void mod_init(or_mod *m, uint64_t q, int logN) {
    int N = 1 << logN;
    m->q = q;
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - q * inv; /* Newton: q^-1 mod 2^64 */
    m->qinv = inv;
    u128 all1 = ~(u128)0;
    u128 fl = all1 / q; /* floor((2^128-1)/q) == floor(2^128/q) because q is odd and > 1 */
    m->bred_hi = (uint64_t)(fl >> 64); m->bred_lo = (uint64_t)fl;
    m->r_mod_q = (uint64_t)((((u128)1) << 64) % q);
    m->r2_mod_q = mulmod(m->r_mod_q, m->r_mod_q, q);
    uint64_t g = or_primitive_root(q);
    uint64_t power = (q - 1) / (2 * (uint64_t)N);
    uint64_t psi = powmod(g, power, q), psi_inv = powmod(g, (q - 1) - power, q);
    uint64_t psi_m = mform(psi, m), psi_inv_m = mform(psi_inv, m);
    m->psi = malloc(sizeof(uint64_t) * (size_t)N); m->psi_inv = malloc(sizeof(uint64_t) * (size_t)N);
    m->psi[0] = m->r_mod_q; m->psi_inv[0] = m->r_mod_q;
    for (int j = 1; j < N; j++) {
        uint32_t prev = bitrev((uint32_t)(j - 1), logN), next = bitrev((uint32_t)j, logN);
        m->psi[next] = mred(m->psi[prev], psi_m, q, m->qinv);
        m->psi_inv[next] = mred(m->psi_inv[prev], psi_inv_m, q, m->qinv);
    }
    m->n_inv = mform(powmod((uint64_t)N, q - 2, q), m);
}

or_ctx *or_ctx_new(int logN, const uint64_t *q, int nq, const uint64_t *p, int np) {
    or_ctx *c = calloc(1, sizeof *c);
    c->logN = logN; c->N = 1 << logN; c->nq = nq; c->np = np;
    c->m = calloc((size_t)(nq + np), sizeof(or_mod));
    for (int i = 0; i < nq; i++) mod_init(&c->m[i], q[i], logN);
    for (int i = 0; i < np; i++) mod_init(&c->m[nq + i], p[i], logN);
    return c;
}
void or_ctx_free(or_ctx *c) {
    if (!c) return;
    for (int i = 0; i < c->nq + c->np; i++) { free(c->m[i].psi); free(c->m[i].psi_inv); }
    free(c->m); free(c);
}
int or_N(const or_ctx *c) { return c->N; }
uint64_t or_modulus(const or_ctx *c, int mod) { return c->m[mod].q; }
const uint64_t *or_psi(const or_ctx *c, int mod) { return c->m[mod].psi; }
const uint64_t *or_psi_inv(const or_ctx *c, int mod) { return c->m[mod].psi_inv; }

/* ---------- NTT ---------- */
/* lattigo ring.NTTLazy (test_run @0x4e97c0): Cooley-Tukey, natural-order input, bit-reversed
 * output, twiddles NttPsi[m+i] in Montgomery form, one MRedConstant per butterfly. Lattigo keeps values
 * below 2^64 by subtracting 4q every other stage; this restatement folds into [0,2q) after every butterfly,
 * which changes no residue class. Output here: [0,2q). */
static void ntt_lazy(const or_mod *m, int N, const uint64_t *in, uint64_t *out) {
    const uint64_t q = m->q, qinv = m->qinv, twoq = 2 * q;
    int t = N >> 1;
    {
        uint64_t F = m->psi[1];
        for (int j = 0; j < t; j++) {
            uint64_t U = in[j], V = mred_lazy(in[j + t], F, q, qinv);
            if (U >= twoq) U -= twoq;
            uint64_t a = U + V, b = U + twoq - V;
            out[j] = a >= twoq ? a - twoq : a;
            out[j + t] = b >= twoq ? b - twoq : b;
        }
    }
    for (int mm = 2; mm < N; mm <<= 1) {
        t >>= 1;
        for (int i = 0; i < mm; i++) {
            uint64_t F = m->psi[mm + i];
            uint64_t *x = out + 2 * i * t, *y = x + t;
            for (int j = 0; j < t; j++) {
                uint64_t U = x[j], V = mred_lazy(y[j], F, q, qinv);
                uint64_t a = U + V, b = U + twoq - V;
                x[j] = a >= twoq ? a - twoq : a;
                y[j] = b >= twoq ? b - twoq : b;
            }
        }
    }
}
/* lattigo ring.NTT = NTTLazy + BRedAdd pass -> canonical [0,q) */
void or_ntt(const or_ctx *c, int mod, const uint64_t *in, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    if (in == out) {
        uint64_t *tmp = malloc(sizeof(uint64_t) * (size_t)c->N);
        memcpy(tmp, in, sizeof(uint64_t) * (size_t)c->N);
        ntt_lazy(m, c->N, tmp, out); free(tmp);
    } else ntt_lazy(m, c->N, in, out);
    for (int j = 0; j < c->N; j++) out[j] = bred_add(out[j], m->q, m->bred_hi);
}
/* lattigo ring.InvNTT (test_run @0x4eb060): Gentleman-Sande, bit-reversed input, natural output, twiddles
 * NttPsiInv[h+i], final multiplication by NttNInv through MRed -> canonical. */
static void intt_core(const or_mod *m, int N, const uint64_t *in, uint64_t *out) {
    const uint64_t q = m->q, qinv = m->qinv, twoq = 2 * q;
    if (in != out) memcpy(out, in, sizeof(uint64_t) * (size_t)N);
    int t = 1;
    for (int mm = N; mm > 1; mm >>= 1) {
        int h = mm >> 1;
        for (int i = 0; i < h; i++) {
            uint64_t F = m->psi_inv[h + i];
            uint64_t *x = out + 2 * i * t, *y = x + t;
            for (int j = 0; j < t; j++) {
                uint64_t U = x[j], V = y[j];
                uint64_t s = U + V; if (s >= twoq) s -= twoq;
                x[j] = s;
                y[j] = mred_lazy(U + twoq - V, F, q, qinv);
            }
        }
        t <<= 1;
    }
}
void or_intt(const or_ctx *c, int mod, const uint64_t *in, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    intt_core(m, c->N, in, out);
    for (int j = 0; j < c->N; j++) out[j] = mred(out[j], m->n_inv, m->q, m->qinv);
}

/* ---------- coefficient-wise ops (lattigo ring.(*Ring).{MForm,MulCoeffsMontgomery,Add,Sub}Lvl) ---------- */
void or_mform(const or_ctx *c, int mod, const uint64_t *in, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    for (int j = 0; j < c->N; j++) out[j] = mred(in[j], m->r2_mod_q, m->q, m->qinv);
}
void or_mul_mont(const or_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    for (int j = 0; j < c->N; j++) out[j] = mred(a[j], b[j], m->q, m->qinv);
}
/* ckks.evaluator.mulRelin ct x pt branch = MFormLvl(pt) then MulCoeffsMontgomeryLvl: a*b mod q */
void or_mul(const or_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    for (int j = 0; j < c->N; j++) out[j] = mred(a[j], mred(b[j], m->r2_mod_q, m->q, m->qinv), m->q, m->qinv);
}
void or_mul_scalar(const or_ctx *c, int mod, const uint64_t *a, uint64_t s, uint64_t *out) {
    const or_mod *m = &c->m[mod];
    uint64_t sm = mform(s % m->q, m);
    for (int j = 0; j < c->N; j++) out[j] = mred(a[j], sm, m->q, m->qinv);
}
void or_add(const or_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    uint64_t q = c->m[mod].q;
    for (int j = 0; j < c->N; j++) out[j] = addmod(a[j], b[j], q);
}
void or_sub(const or_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    uint64_t q = c->m[mod].q;
    for (int j = 0; j < c->N; j++) out[j] = submod(a[j], b[j], q);
}

/* lattigo ring.PermuteNTTIndex (test_run @0x4e2c80): idx[i] = bitrev(((galEl*(2*bitrev(i)+1) mod 2N) - 1)/2) */
void or_permute_index(int logN, uint64_t galEl, uint32_t *idx) {
    uint64_t N = 1ull << logN, mask = 2 * N - 1;
    for (uint64_t i = 0; i < N; i++) {
        uint64_t t1 = 2 * (uint64_t)bitrev((uint32_t)i, logN) + 1;
        uint64_t t2 = ((galEl * t1 & mask) - 1) >> 1;
        idx[i] = bitrev((uint32_t)t2, logN);
    }
}
/* lattigo ring.PermuteNTTWithIndexLvl: gather, no sign changes in the NTT domain */
void or_permute(int N, const uint32_t *idx, const uint64_t *in, uint64_t *out) {
    for (int i = 0; i < N; i++) out[i] = in[idx[i]];
}

/* ---------- MultByConst / Rescale (lattigo ckks.(*evaluator).{getConstAndScale,MultByConst,Rescale},
 * ckks.scaleUpExact; test_run @0x51d620, @0x5320a0, @0x522440) ---------- */
uint64_t or_const_for(double constant, double q_level_f, uint64_t q, double *scale_mult) {
    double scale = 1.0;
    if (constant != 0) {
        double frac = constant - (double)(int64_t)constant;   /* cvttsd2si / cvtsi2sd / subsd */
        if (frac != 0) scale = q_level_f;
    }
    if (scale_mult) *scale_mult = scale;
    /* scaleUpExact: big.NewFloat(n*value) [53-bit], Add 0.5 [53-bit, RNE], Int() truncates, Mod q; q-res if negative */
    int neg = constant < 0;
    double x = neg ? (-scale * constant) : (scale * constant);
    x = x + 0.5;
    /* x may exceed 2^64 in general (big.Int path); on the conv path it is < 2^53 */
    uint64_t res;
    if (x < 18446744073709551616.0) res = (uint64_t)x % q;
    else { /* exact big-integer value of the double, reduced mod q */
        int e; double mant = frexp(x, &e); /* x = mant * 2^e, mant in [0.5,1) */
        uint64_t mi = (uint64_t)ldexp(mant, 53); int sh = e - 53;
        uint64_t r = mi % q;
        for (int i = 0; i < sh; i++) r = addmod(r, r, q);
        res = r;
    }
    if (neg) res = q - res;
    return res;
}
int or_rescale_drops(const or_ctx *c, int level, double scale, double min_scale, double *scale_out) {
    int n = 0;
    /* for ctOut.Scale/float64(Q[level-n]) >= minScale/2 && level-n >= 0 { Scale /= ...; n++ }  (upstream v2.2.0;
     * test_run @0x52253e-0x522572). The loop never runs with level-n < 0 because a drop to below level 0 would
     * index Q[-1]; Lattigo guards with Level()==0 -> error before the loop, and on this path it stops at level 0. */
    while (level - n > 0 && scale / (double)c->m[level - n].q >= min_scale / 2) {
        scale /= (double)c->m[level - n].q; n++;
    }
    if (scale_out) *scale_out = scale;
    return n;
}
/* lattigo ring.divRoundByLastModulusNTT (test_run @0x4f2b20):
 * t = InvNTT(x_L); h = (q_L-1)>>1; t = CRed(t+h, q_L); for i<L: u = NTTLazy_i(t + (q_i - h mod q_i));
 * x_i = MRed(u + 2q_i - x_i, -q_L^-1 * 2^64 mod q_i). */
void or_div_round_last_ntt(const or_ctx *c, int level, const uint64_t *x, uint64_t *out) {
    int N = c->N; const or_mod *mL = &c->m[level];
    uint64_t qL = mL->q, h = (qL - 1) >> 1;
    uint64_t *t = malloc(sizeof(uint64_t) * (size_t)N), *u = malloc(sizeof(uint64_t) * (size_t)N),
             *v = malloc(sizeof(uint64_t) * (size_t)N);
    or_intt(c, level, x + (size_t)level * (size_t)N, t);
    for (int j = 0; j < N; j++) { uint64_t s = t[j] + h; if (s >= qL) s -= qL; t[j] = s; }
    for (int i = 0; i < level; i++) {
        const or_mod *m = &c->m[i];
        uint64_t qi = m->q, neg_h = qi - (h % qi);
        /* -q_L^-1 in Montgomery form */
        uint64_t qlinv = powmod(qL % qi, qi - 2, qi);
        uint64_t k = mform(qi - qlinv, m);
        for (int j = 0; j < N; j++) v[j] = t[j] + neg_h;     /* unreduced, < 2^64 (lattigo: AddScalar lazily) */
        for (int j = 0; j < N; j++) v[j] = bred_add(v[j], qi, m->bred_hi);
        ntt_lazy(m, N, v, u);
        const uint64_t *xi = x + (size_t)i * (size_t)N; uint64_t *oi = out + (size_t)i * (size_t)N;
        for (int j = 0; j < N; j++) oi[j] = mred(u[j] + 2 * qi - xi[j], k, qi, m->qinv);
    }
    free(t); free(u); free(v);
}

/* the OpenMP team of the general-level key switch's row loops (test infrastructure only; bench.py's cpu_baseline times or_conv_then_pack, which is single-threaded).
 * Set by the loader (tests/oracle_lib.py) through this call, NOT through OMP_NUM_THREADS: an environment variable would be inherited by every subprocess a test starts. */
#ifdef _OPENMP
#include <omp.h>
void or_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void or_set_threads(int n) { (void)n; }
#endif

/* ---------- key switching (lattigo rlwe.(*KeySwitcher).SwitchKeysInPlace[NoModDown], DecomposeSingleNTT,
 * ring.(*Decomposer).DecomposeAndSplit, ring.(*FastBasisExtender).ModDownSplitNTTPQ, ring.modUpExact;
 * test_run @0x4fdd40, @0x4fe660, @0x4fe260, @0x4e6400, @0x4e4c40, @0x4e5700) ---------- */
/* single P prime -> one Q prime: y = MRed(x, MForm((P/p)^-1 = 1)) = x mod p; v = uint64(float64(y)/float64(p))
 * (fp64 division, truncation; 1 only when rounding pushes the quotient to 1.0); out = y - v*P mod q. */
uint64_t or_modup_1p(uint64_t y, uint64_t p, uint64_t q) {
    double vf = (double)y / (double)p;
    uint64_t v = (uint64_t)vf;
    uint64_t r = y % q;
    if (v) r = submod(r, mulmod(v % q, p % q, q), q);
    return r;
}
void or_keyswitch_l0(const or_ctx *c, const uint64_t *c1, const uint64_t *evk_b_q, const uint64_t *evk_a_q,
                     const uint64_t *evk_b_p, const uint64_t *evk_a_p, uint64_t *d0, uint64_t *d1) {
    int N = c->N; const int Q0 = 0, P = c->nq;
    const or_mod *mq = &c->m[Q0], *mp = &c->m[P];
    uint64_t *cc = malloc(sizeof(uint64_t) * (size_t)N), *cp = malloc(sizeof(uint64_t) * (size_t)N);
    uint64_t *acc = malloc(sizeof(uint64_t) * (size_t)N), *ext = malloc(sizeof(uint64_t) * (size_t)N);
    /* cxInvNTT = InvNTT(cx) (canonical); digit 0 is the single limb Q0: its residues are copied unreduced into
     * the P limb (c < Q0 < P) and NTTLazy'd there; the Q0 limb reuses the NTT-domain input (DecomposeAndSplit). */
    or_intt(c, Q0, c1, cc);
    ntt_lazy(mp, N, cc, cp);
    uint64_t pinv_q = powmod(mp->q % mq->q, mq->q - 2, mq->q);
    uint64_t pinv_m = mform(pinv_q, mq);
    const uint64_t *eq[2] = {evk_b_q, evk_a_q}, *ep[2] = {evk_b_p, evk_a_p};
    uint64_t *dd[2] = {d0, d1};
    for (int k = 0; k < 2; k++) {
        /* P part: acc_P = MRedConstant(evk_P, c_P) then Reduce; InvNTTLazy_P; y = value mod P */
        for (int j = 0; j < N; j++) acc[j] = mred(ep[k][j], cp[j], mp->q, mp->qinv);
        intt_core(mp, N, acc, acc);
        for (int j = 0; j < N; j++) {
            uint64_t y = mred(acc[j], mp->n_inv, mp->q, mp->qinv);     /* InvNTTLazy includes N^-1; canonical here */
            ext[j] = or_modup_1p(y, mp->q, mq->q);
        }
        ntt_lazy(mq, N, ext, ext);
        /* Q part: acc_Q = evk_Q (*) c1, then (acc_Q - NTT(ext)) * P^-1 */
        for (int j = 0; j < N; j++) {
            uint64_t aq = mred(eq[k][j], c1[j], mq->q, mq->qinv);
            dd[k][j] = mred(aq + 2 * mq->q - ext[j], pinv_m, mq->q, mq->qinv);
        }
    }
    free(cc); free(cp); free(acc); free(ext);
}

/* lattigo ring.reconstructRNS + ring.multSum (test_run @0x4e79b0 region; SURVEY 8(a)-R "modUpExact" row):
 *   y_i = x_i * (S/s_i)^-1 mod s_i (canonical);  v = uint64( sum_i float64(y_i)/float64(s_i) )  [fp64, in limb order];
 *   result = sum_i y_i * (S/s_i mod t) - v * (S mod t)  mod t.      S = prod s_i. */
/* constants of one extension {src} -> t: inv[i] = (S/s_i)^-1 mod s_i, hat[i] = (S/s_i) mod t, S mod t */
typedef struct { int n; uint64_t src[16], inv[16], hat[16], S, t; } bx_pre;
static void bx_prepare(bx_pre *b, const uint64_t *src, int n, uint64_t t) {
    b->n = n; b->t = t; b->S = 1;
    for (int i = 0; i < n; i++) {
        uint64_t si = src[i], hat_mod_si = 1, hat_mod_t = 1;
        for (int j = 0; j < n; j++) if (j != i) { hat_mod_si = mulmod(hat_mod_si, src[j] % si, si); hat_mod_t = mulmod(hat_mod_t, src[j] % t, t); }
        b->src[i] = si; b->inv[i] = powmod(hat_mod_si, si - 2, si); b->hat[i] = hat_mod_t;
        b->S = mulmod(b->S, si % t, t);
    }
}
static inline uint64_t bx_apply(const bx_pre *b, const uint64_t *x) {
    double vi = 0.0; uint64_t acc = 0;
    for (int i = 0; i < b->n; i++) {
        const uint64_t si = b->src[i], y = mulmod(x[i] % si, b->inv[i], si);
        vi += (double)y / (double)si;
        acc = addmod(acc, mulmod(y % b->t, b->hat[i], b->t), b->t);
    }
    const uint64_t v = (uint64_t)vi;
    return submod(acc, mulmod(v % b->t, b->S, b->t), b->t);
}
uint64_t or_basis_extend(const uint64_t *x, const uint64_t *src, int n, uint64_t t) {
    bx_pre b; bx_prepare(&b, src, n, t);
    return bx_apply(&b, x);
}

/* general rlwe.(*KeySwitcher).SwitchKeysInPlace (NTT-domain input); see oracle.h. Its two halves are callable on their own:
 * or_keyswitch_qp  = rlwe.(*KeySwitcher).SwitchKeysInPlaceNoModDown / DecomposeNTT + KeyswitchHoistedNoModDown (test_run @4fe660, @4fdf80,
 *                    @4ff060): digit decomposition, basis extension of every digit to all level+1+alpha limbs, inner product with the key:
 *                    acc[k][limb][N], limbs = Q_0..Q_level then P_0..P_(alpha-1), canonical residues, NTT domain;
 * or_mod_down      = ring.(*FastBasisExtender).ModDownSplitNTTPQ (@4e4c40) on ONE polynomial in that layout: InvNTT of the P limbs, extension
 *                    {P} -> each Q limb, (x_Q - NTT(ext)) * P^-1. */
/* rlwe.(*KeySwitcher).DecomposeNTT: digits[d][T][N] = the d-th digit of cx extended to limb T (Q_0..Q_level, then the P limbs), NTT domain */
void or_keyswitch_decompose(const or_ctx *c, int level, const uint64_t *cx, uint64_t *digits) {
    const int N = c->N, alpha = c->np, nl = level + 1, nt = nl + alpha;          /* nt = limbs an evk row set covers */
    const int beta = (nl + alpha - 1) / alpha;
    const size_t n = (size_t)N;
    uint64_t *coef = malloc(sizeof(uint64_t) * n * (size_t)nl);                   /* cxInvNTT */
    /* The rows of the general-level functions (this one, or_keyswitch_mac, or_mod_down) are independent and run on OpenMP threads when the library is built with -fopenmp
     * (oracle/Makefile): the Python replays of the reference's bootstrapping chains spend their time here. The hot path's functions - what bench.py times as cpu_baseline -
     * carry no pragma and stay on one thread. */
#pragma omp parallel for schedule(dynamic)
    for (int l = 0; l < nl; l++) or_intt(c, l, cx + (size_t)l * n, coef + (size_t)l * n);
    for (int d = 0; d < beta; d++) {
        const int lo = d * alpha, hi = (d + 1) * alpha < nl ? (d + 1) * alpha : nl, nd = hi - lo;
        uint64_t src[16]; for (int i = 0; i < nd; i++) src[i] = c->m[lo + i].q;
#pragma omp parallel for schedule(dynamic)
        for (int T = 0; T < nt; T++) {
            uint64_t *tmp = malloc(sizeof(uint64_t) * n);
            const int mod = T < nl ? T : c->nq + (T - nl);                         /* ctx modulus index of target limb */
            const or_mod *m = &c->m[mod];
            uint64_t *c2 = digits + ((size_t)d * (size_t)nt + (size_t)T) * n;
            if (T >= lo && T < hi) memcpy(c2, cx + (size_t)T * n, sizeof(uint64_t) * n);   /* the digit's own limbs: NTT input reused */
            else {
                if (nd == 1) for (int j = 0; j < N; j++) tmp[j] = coef[(size_t)lo * n + (size_t)j] % m->q;   /* copied residues */
                else {
                    bx_pre bx; bx_prepare(&bx, src, nd, m->q);
                    for (int j = 0; j < N; j++) {
                        uint64_t x[16]; for (int i = 0; i < nd; i++) x[i] = coef[(size_t)(lo + i) * n + (size_t)j];
                        tmp[j] = bx_apply(&bx, x);
                    }
                }
                or_ntt(c, mod, tmp, c2);
            }
            free(tmp);
        }
    }
    free(coef);
}
/* rlwe.(*KeySwitcher).KeyswitchHoistedNoModDown: the inner product of a decomposition with one key */
void or_keyswitch_mac(const or_ctx *c, int level, const uint64_t *digits, const uint64_t *evk, uint64_t *acc) {
    const int N = c->N, alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha;
    const size_t n = (size_t)N;
    memset(acc, 0, sizeof(uint64_t) * n * (size_t)nt * 2);                        /* [k][limb][N] */
#pragma omp parallel for schedule(dynamic)
    for (int T = 0; T < nt; T++) for (int d = 0; d < beta; d++) {                 /* (modular sums: the order of the digits does not matter) */
        const or_mod *m = &c->m[T < nl ? T : c->nq + (T - nl)];
        const uint64_t *c2 = digits + ((size_t)d * (size_t)nt + (size_t)T) * n;
        for (int k = 0; k < 2; k++) {
            const uint64_t *e = evk + (((size_t)d * 2 + (size_t)k) * (size_t)nt + (size_t)T) * n;
            uint64_t *a = acc + ((size_t)k * (size_t)nt + (size_t)T) * n;
            for (int j = 0; j < N; j++) a[j] = addmod(a[j], mred(e[j], c2[j], m->q, m->qinv), m->q);
        }
    }
}
void or_keyswitch_qp(const or_ctx *c, int level, const uint64_t *cx, const uint64_t *evk, uint64_t *acc) {
    const int nt = level + 1 + c->np, beta = (level + 1 + c->np - 1) / c->np;
    uint64_t *digits = malloc(sizeof(uint64_t) * (size_t)c->N * (size_t)nt * (size_t)beta);
    or_keyswitch_decompose(c, level, cx, digits);
    or_keyswitch_mac(c, level, digits, evk, acc);
    free(digits);
}
void or_mod_down(const or_ctx *c, int level, const uint64_t *x_qp, uint64_t *out) {
    const int N = c->N, alpha = c->np, nl = level + 1;
    const size_t n = (size_t)N;
    uint64_t psrc[16]; for (int j = 0; j < alpha; j++) psrc[j] = c->m[c->nq + j].q;
    uint64_t *pc = malloc(sizeof(uint64_t) * n * (size_t)alpha);
    for (int j = 0; j < alpha; j++) or_intt(c, c->nq + j, x_qp + (size_t)(nl + j) * n, pc + (size_t)j * n);
#pragma omp parallel for schedule(dynamic)
    for (int l = 0; l < nl; l++) {
        uint64_t *tmp = malloc(sizeof(uint64_t) * n);
        const or_mod *m = &c->m[l];
        uint64_t pinv = 1; for (int j = 0; j < alpha; j++) pinv = mulmod(pinv, psrc[j] % m->q, m->q);
        pinv = powmod(pinv, m->q - 2, m->q);
        bx_pre bx; bx_prepare(&bx, psrc, alpha, m->q);
        for (int j = 0; j < N; j++) { uint64_t x[16]; for (int i = 0; i < alpha; i++) x[i] = pc[(size_t)i * n + (size_t)j]; tmp[j] = bx_apply(&bx, x); }
        or_ntt(c, l, tmp, tmp);
        const uint64_t *a = x_qp + (size_t)l * n;
        for (int j = 0; j < N; j++) out[(size_t)l * n + (size_t)j] = mulmod(submod(a[j] % m->q, tmp[j], m->q), pinv, m->q);
        free(tmp);
    }
    free(pc);
}
void or_keyswitch(const or_ctx *c, int level, const uint64_t *cx, const uint64_t *evk, uint64_t *d0, uint64_t *d1) {
    const size_t n = (size_t)c->N, nt = (size_t)(level + 1 + c->np);
    uint64_t *acc = malloc(sizeof(uint64_t) * n * nt * 2);
    or_keyswitch_qp(c, level, cx, evk, acc);
    or_mod_down(c, level, acc, d0);
    or_mod_down(c, level, acc + nt * n, d1);
    free(acc);
}

/* lattigo ckks.(*evaluator).RotateGal -> permuteNTT (test_run @0x5245a0, @0x5248c0): key-switch c1, add c0 to
 * the first component, permute both. */
void or_rotate_gal_l0(const or_ctx *c, const uint64_t *c0, const uint64_t *c1, const uint32_t *perm_idx,
                      const uint64_t *evk_b_q, const uint64_t *evk_a_q, const uint64_t *evk_b_p,
                      const uint64_t *evk_a_p, uint64_t *o0, uint64_t *o1) {
    int N = c->N;
    uint64_t *d0 = malloc(sizeof(uint64_t) * (size_t)N), *d1 = malloc(sizeof(uint64_t) * (size_t)N);
    or_keyswitch_l0(c, c1, evk_b_q, evk_a_q, evk_b_p, evk_a_p, d0, d1);
    or_add(c, 0, d0, c0, d0);
    or_permute(N, perm_idx, d0, o0);
    or_permute(N, perm_idx, d1, o1);
    free(d0); free(d1);
}

/* conv.go:527-528: MulNew(ct_in, pl_ker[i]) ; SetScale(ct, out_scale/(max_ob/norm)).
 * SetScale (test_run @0x521b80) = MultByConst(ct, scale/ct.Scale) ; Rescale(ct, scale) ; ct.Scale = scale.
 * Here: level 1 -> one drop -> level 0. */
void or_mul_setscale(const or_ctx *c, const uint64_t *ct_in, const uint64_t *pl_ker, const uint64_t cst[2],
                     uint64_t *ct_out) {
    int N = c->N; size_t n = (size_t)N;
    uint64_t *a = malloc(sizeof(uint64_t) * 2 * n);
    for (int p = 0; p < 2; p++) {
        for (int l = 0; l < 2; l++) {
            or_mul(c, l, ct_in + ((size_t)p * 2 + (size_t)l) * n, pl_ker + (size_t)l * n, a + (size_t)l * n);
            or_mul_scalar(c, l, a + (size_t)l * n, cst[l], a + (size_t)l * n);
        }
        or_div_round_last_ntt(c, 1, a, ct_out + (size_t)p * n);
    }
    free(a);
}

/* conv.go:522-546 conv_then_pack + conv.go:266-300 pack_ctxts + eval.go:258 bias add */
double or_conv_then_pack(const or_ctx *c, const uint64_t *ct_in, double ct_scale, const uint64_t *pl_ker,
                         double ker_scale, const uint64_t *idx_pt, const uint64_t *evk, int max_ob, int norm,
                         double out_scale, const uint64_t *bias, uint64_t *ct_out) {
    int N = c->N; size_t n = (size_t)N;
    uint64_t *cts = malloc(sizeof(uint64_t) * 2 * n * (size_t)max_ob);
    /* loop A (conv.go:525-531) */
    double target = out_scale / (double)(max_ob / norm);
    double prod_scale = ct_scale * ker_scale;
    double constant = target / prod_scale;
    uint64_t cst[2]; double smul = 1;
    for (int l = 0; l < 2; l++) cst[l] = or_const_for(constant, (double)c->m[1].q, c->m[l].q, &smul);
    double sc_after;
    int drops = or_rescale_drops(c, 1, prod_scale * smul, target, &sc_after);
    if (drops != 1) { free(cts); return -1.0; }
    for (int i = 0; i < max_ob; i++)
        if (i % norm == 0) or_mul_setscale(c, ct_in, pl_ker + (size_t)i * 2 * n, cst, cts + (size_t)i * 2 * n);
    /* pack_ctxts (conv.go:266-300): scale bookkeeping (conv.go:274) then the tree */
    int real_cnum = max_ob / norm;
    double scale = target * (double)real_cnum;
    int step = max_ob / 2, logStep = 0;
    for (int i = step; i > 1; i /= 2) logStep++;
    int j = c->logN - logStep;
    uint64_t *t1 = malloc(sizeof(uint64_t) * 2 * n), *t2 = malloc(sizeof(uint64_t) * 2 * n), *r = malloc(sizeof(uint64_t) * 2 * n);
    uint32_t *perm = malloc(sizeof(uint32_t) * n);
    while (step >= norm && step >= 1) {
        or_permute_index(c->logN, (1ull << j) + 1, perm);
        const uint64_t *I = idx_pt + (size_t)logStep * n;
        const uint64_t *k4 = evk + (size_t)(j - 1) * 4 * n;
        for (int i = 0; i < step; i += norm) {
            uint64_t *x = cts + (size_t)(i + step) * 2 * n, *y = cts + (size_t)i * 2 * n;
            for (int p = 0; p < 2; p++) {
                or_mul(c, 0, x + (size_t)p * n, I, t1 + (size_t)p * n);                 /* conv.go:288 */
                or_sub(c, 0, y + (size_t)p * n, t1 + (size_t)p * n, t2 + (size_t)p * n); /* conv.go:289 */
                or_add(c, 0, y + (size_t)p * n, t1 + (size_t)p * n, t1 + (size_t)p * n); /* conv.go:290 */
            }
            or_rotate_gal_l0(c, t2, t2 + n, perm, k4, k4 + n, k4 + 2 * n, k4 + 3 * n, r, r + n); /* conv.go:291 */
            for (int p = 0; p < 2; p++) or_add(c, 0, t1 + (size_t)p * n, r + (size_t)p * n, y + (size_t)p * n); /* :292 */
        }
        step /= 2; logStep--; j++;
    }
    memcpy(ct_out, cts, sizeof(uint64_t) * 2 * n);
    if (bias) or_add(c, 0, ct_out, bias, ct_out);                                        /* eval.go:258 */
    free(t1); free(t2); free(r); free(perm); free(cts);
    return scale;
}

/* ---------- encoding (lattigo ckks.(*encoderComplex128).EncodeCoeffs @0x518ba0 -> scaleUpVecExact @0x532380) ---------- */
void or_encode_coeffs(const or_ctx *c, const double *v, int n, double scale, const int *mods, int nmods, uint64_t *out) {
    size_t N = (size_t)c->N;
    for (int k = 0; k < nmods; k++) memset(out + (size_t)k * N, 0, sizeof(uint64_t) * N);
    for (int i = 0; i < n; i++) {
        double val = v[i];
        int neg = val < 0;
        double x = neg ? (-scale * val) : (scale * val);   /* plain f64 product */
        if (x > 1.8446744073709552e+19) {                  /* big.Float path: +0.5, Int, Mod */
            double y = x + 0.5; int e; double mant = frexp(y, &e);
            uint64_t mi = (uint64_t)ldexp(mant, 53); int sh = e - 53;
            for (int k = 0; k < nmods; k++) {
                uint64_t q = c->m[mods[k]].q, r = mi % q;
                for (int s = 0; s < sh; s++) r = addmod(r, r, q);
                out[(size_t)k * N + (size_t)i] = neg ? q - r : r;
            }
        } else {
            uint64_t xi = (uint64_t)(x + 0.5);
            for (int k = 0; k < nmods; k++) {
                const or_mod *m = &c->m[mods[k]];
                uint64_t r = xi % m->q;
                out[(size_t)k * N + (size_t)i] = neg ? m->q - r : r;   /* q (non-canonical) when r == 0, as upstream */
            }
        }
    }
}

/* main.go:1007-1042 prep_Input, trans == false branch */
void or_prep_input(const double *input, int raw_in_wid, int in_wid, int N, int norm, double *out) {
    int batch = N / (in_wid * in_wid), k = 0;
    memset(out, 0, sizeof(double) * (size_t)N);
    for (int i = 0; i < in_wid; i++)
        for (int j = 0; j < in_wid; j++)
            for (int b = 0; b < batch / norm; b++)
                if (i < raw_in_wid && j < raw_in_wid) out[i * in_wid * batch + j * batch + b * norm] = input[k++];
}
/* conv.go:184-202 reshape_ker, trans == false: ker_out[i][j*k_sz+k] = ker_in[i + j*out_batch + k*out_batch*in_batch] */
void or_reshape_ker(const double *ker_in, int len, int k_sz, int out_batch, double *ker_out) {
    int in_batch = len / (k_sz * out_batch);
    for (int i = 0; i < out_batch; i++)
        for (int j = 0; j < in_batch; j++)
            for (int k = 0; k < k_sz; k++)
                ker_out[(size_t)i * (size_t)(k_sz * in_batch) + (size_t)(j * k_sz + k)] =
                    ker_in[i + j * out_batch + k * out_batch * in_batch];
}
/* conv.go:206-237 encode_ker_final */
void or_encode_ker_final(const double *ker_rs, int row_len, int pos, int i, int in_wid, int in_batch, int ker_wid,
                         double *output) {
    int vec_size = in_wid * in_wid * in_batch, k_sz = ker_wid * ker_wid;
    int bias = pos * ker_wid * ker_wid * in_batch;
    memset(output, 0, sizeof(double) * (size_t)vec_size);
    const double *row = ker_rs + (size_t)i * (size_t)row_len;
    for (int j = 0; j < in_batch; j++)
        for (int k = 0; k < k_sz; k++)
            output[(in_wid * (k / ker_wid) + k % ker_wid) * in_batch + j] = row[(in_batch - 1 - j) * k_sz + (k_sz - 1 - k) + bias];
    int adj = (in_batch - 1) + in_batch * (in_wid + 1) * (ker_wid - 1) / 2;
    double *tmp = malloc(sizeof(double) * (size_t)adj);
    for (int t = 0; t < adj; t++) { tmp[t] = output[vec_size - adj + t]; output[vec_size - adj + t] = -output[t]; }
    for (int t = 0; t < vec_size - 2 * adj; t++) output[t] = output[t + adj];
    for (int t = 0; t < adj; t++) output[t + vec_size - 2 * adj] = tmp[t];
    free(tmp);
}
/* conv.go:487-515 prep_Ker without the encoder calls */
void or_prep_ker_coeffs(const double *ker_in, int ker_len, const double *bn_a, int N, int in_wid, int ker_wid,
                        int real_ib, int real_ob, int norm, double *out) {
    int max_bat = N / (in_wid * in_wid), ker_size = ker_wid * ker_wid;
    int in_batch = ker_len / (ker_size * real_ob);
    double *ker_rs = malloc(sizeof(double) * (size_t)real_ob * (size_t)(ker_size * in_batch));
    or_reshape_ker(ker_in, ker_len, ker_size, real_ob, ker_rs);
    for (int i = 0; i < real_ob; i++)
        for (int j = 0; j < ker_size * in_batch; j++) ker_rs[(size_t)i * (size_t)(ker_size * in_batch) + (size_t)j] *= bn_a[i];
    int row_len = max_bat * ker_size;
    double *maxk = calloc((size_t)max_bat * (size_t)row_len, sizeof(double));
    for (int i = 0; i < real_ob; i++)
        for (int j = 0; j < real_ib; j++)
            for (int k = 0; k < ker_size; k++)
                maxk[(size_t)(norm * i) * (size_t)row_len + (size_t)(norm * j * ker_size + k)] =
                    ker_rs[(size_t)i * (size_t)(ker_size * in_batch) + (size_t)(j * ker_size + k)];
    for (int i = 0; i < max_bat; i++)
        or_encode_ker_final(maxk, row_len, 0, i, in_wid, max_bat, ker_wid, out + (size_t)i * (size_t)N);
    free(ker_rs); free(maxk);
}
/* main.go:1057-1070 post_process */
void or_post_process(const double *in_cfs, int len, int raw_in_wid, int in_wid, double *out) {
    int batch = len / (in_wid * in_wid);
    for (int i = 0; i < raw_in_wid; i++)
        for (int j = 0; j < raw_in_wid; j++)
            for (int b = 0; b < batch; b++)
                out[i * raw_in_wid * batch + batch * j + b] = in_cfs[i * in_wid * batch + batch * j + b];
}
/* eval.go:233-238 */
void or_bias_coeffs(const double *bn_b, int real_ob, int N, int in_wid, int norm, double *out) {
    int max_batch = N / (in_wid * in_wid);
    memset(out, 0, sizeof(double) * (size_t)N);
    for (int i = 0; i < real_ob; i++)
        for (int j = 0; j < in_wid * in_wid; j++) out[norm * i + j * max_batch] = bn_b[i];
}

/* ---------- harness-only crypto ---------- */
static inline uint64_t sm64(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void or_fill_seeded(uint64_t seed, uint64_t q, int n, uint64_t *out) {
    for (int j = 0; j < n; j++) out[j] = sm64(seed, (uint64_t)j) % q;
}
/* main.go:410 GenKeyPairSparse(h): exactly h non-zero coefficients in {-1,+1} */
void or_gen_sk(const or_ctx *c, uint64_t seed, int h, int64_t *sk) {
    int N = c->N; memset(sk, 0, sizeof(int64_t) * (size_t)N);
    uint64_t ctr = 0; int placed = 0;
    while (placed < h) {
        uint64_t r = sm64(seed, ctr++); int pos = (int)(r % (uint64_t)N);
        if (sk[pos] == 0) { sk[pos] = (r >> 40) & 1 ? 1 : -1; placed++; }
    }
}
static void signed_rows(const or_ctx *c, const int64_t *v, int mod, uint64_t *out) {
    uint64_t q = c->m[mod].q;
    for (int j = 0; j < c->N; j++) out[j] = v[j] >= 0 ? (uint64_t)v[j] % q : q - ((uint64_t)(-v[j]) % q);
}
void or_sk_rows(const or_ctx *c, const int64_t *sk, int mod, uint64_t *out_ntt) {
    signed_rows(c, sk, mod, out_ntt); or_ntt(c, mod, out_ntt, out_ntt);
}
/* discrete Gaussian, sigma = 3.2 (main.go:421 rlwe.DefaultSigma), truncated at 6 sigma */
static void gauss(uint64_t seed, int N, int64_t *e) {
    for (int j = 0; j < N; j++) {
        for (uint64_t k = 0;; k++) {
            double u1 = ((double)(sm64(seed, (uint64_t)j * 64 + 2 * k) >> 11) + 1.0) / 9007199254740993.0;
            double u2 = (double)(sm64(seed, (uint64_t)j * 64 + 2 * k + 1) >> 11) / 9007199254740992.0;
            double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * 3.2;
            if (fabs(g) <= 19.2) { e[j] = (int64_t)llround(g); break; }
        }
    }
}
/* lattigo rlwe.(*keyGenerator).genrotKey/newSwitchingKey (test_run @0x4fb1c0, @0x4fb320), digit 0 at level 0:
 * skOut = sigma_{galEl^-1}(sk); b = -a*skOut + e + P*sk  (mod Q0 and mod P; P*sk vanishes mod P); stored NTT+Montgomery.
 * Self-consistency requirement only (SURVEY 8(a)-R last rows): keyswitch(c1, evk_g) then Permute_g decrypts to sigma_g(m). */
void or_gen_galois_key_l0(const or_ctx *c, const int64_t *sk, uint64_t galEl, uint64_t seed, uint64_t *evk4) {
    int N = c->N; size_t n = (size_t)N; int P = c->nq;
    uint64_t twoN = 2 * (uint64_t)N, ginv = 1, g = galEl % twoN;
    for (uint64_t e = twoN - 1, b = g; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN; /* ModExp(galEl, 2N-1, 2N) */
    /* sigma_{ginv}(sk) in the coefficient domain: X^i -> X^(i*ginv mod 2N), sign flip when >= N */
    int64_t *sko = calloc(n, sizeof(int64_t)), *e = malloc(sizeof(int64_t) * n);
    for (int i = 0; i < N; i++) {
        uint64_t t = ((uint64_t)i * ginv) % twoN;
        if (t < (uint64_t)N) sko[t] = sk[i]; else sko[t - (uint64_t)N] = -sk[i];
    }
    gauss(seed ^ 0xE44E44ull, N, e);
    uint64_t *s_in = malloc(sizeof(uint64_t) * n), *s_out = malloc(sizeof(uint64_t) * n), *en = malloc(sizeof(uint64_t) * n);
    int mods[2] = {0, P};
    for (int w = 0; w < 2; w++) {
        int mod = mods[w]; const or_mod *m = &c->m[mod];
        uint64_t *b = evk4 + (size_t)(w * 2) * n, *a = evk4 + (size_t)(w * 2 + 1) * n;
        or_fill_seeded(seed + 0x1000 + (uint64_t)w, m->q, N, a);                 /* uniform a, already "NTT domain" */
        or_sk_rows(c, sk, mod, s_in); or_sk_rows(c, sko, mod, s_out);
        signed_rows(c, e, mod, en); or_ntt(c, mod, en, en);
        uint64_t pmod = (w == 0) ? c->m[P].q % m->q : 0;
        for (int j = 0; j < N; j++) {
            uint64_t v = submod(en[j], mulmod(a[j], s_out[j], m->q), m->q);
            v = addmod(v, mulmod(pmod, s_in[j], m->q), m->q);
            b[j] = mform(v, m); a[j] = mform(a[j], m);                           /* stored in Montgomery form */
        }
    }
    free(sko); free(e); free(s_in); free(s_out); free(en);
}
/* general version of the above (any level, np special primes, beta digits of np limbs): digit d carries P*s on the Q
 * limbs that belong to digit d only ((Q/Q_d)*[(Q/Q_d)^-1]_{Q_d} is 1 on those limbs and 0 on the others) */
void or_gen_swk(const or_ctx *c, const int64_t *sk, uint64_t galEl, int level, uint64_t seed, uint64_t *rows) {
    const int N = c->N, alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha; const size_t n = (size_t)N;
    const int relin = galEl == 0;                 /* galEl 0: relinearisation key, s_in = s^2, s_out = s */
    uint64_t twoN = 2 * (uint64_t)N, ginv = 1, g = galEl % twoN;
    for (uint64_t e = twoN - 1, b = g; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN;
    int64_t *sko = calloc(n, sizeof(int64_t)), *e = malloc(sizeof(int64_t) * n);
    if (relin) memcpy(sko, sk, sizeof(int64_t) * n);
    else for (int i = 0; i < N; i++) { uint64_t t = ((uint64_t)i * ginv) % twoN; if (t < (uint64_t)N) sko[t] = sk[i]; else sko[t - (uint64_t)N] = -sk[i]; }
    uint64_t *s_in = malloc(sizeof(uint64_t) * n), *s_out = malloc(sizeof(uint64_t) * n), *en = malloc(sizeof(uint64_t) * n);
    for (int d = 0; d < beta; d++) {
        gauss(seed ^ (0xE44E44ull + (uint64_t)d * 7919), N, e);
        for (int T = 0; T < nt; T++) {
            const int mod = T < nl ? T : c->nq + (T - nl); const or_mod *m = &c->m[mod];
            uint64_t *b = rows + (((size_t)d * 2 + 0) * (size_t)nt + (size_t)T) * n, *a = rows + (((size_t)d * 2 + 1) * (size_t)nt + (size_t)T) * n;
            or_fill_seeded(seed + 0x1000 + (uint64_t)(d * 64 + T), m->q, N, a);
            or_sk_rows(c, sk, mod, s_in); or_sk_rows(c, sko, mod, s_out);
            if (relin) for (int j = 0; j < N; j++) s_in[j] = mulmod(s_in[j], s_in[j], m->q);
            signed_rows(c, e, mod, en); or_ntt(c, mod, en, en);
            uint64_t pmod = 0;
            if (T < nl && T >= d * alpha && T < (d + 1) * alpha) { pmod = 1; for (int j = 0; j < alpha; j++) pmod = mulmod(pmod, c->m[c->nq + j].q % m->q, m->q); }
            for (int j = 0; j < N; j++) {
                uint64_t v = submod(en[j], mulmod(a[j], s_out[j], m->q), m->q);
                v = addmod(v, mulmod(pmod, s_in[j], m->q), m->q);
                b[j] = mform(v, m); a[j] = mform(a[j], m);
            }
        }
    }
    free(sko); free(e); free(s_in); free(s_out); free(en);
}
/* lattigo rlwe.(*skEncryptor).encrypt: c1 uniform, c0 = -c1*s + e + m (all NTT) */
void or_encrypt(const or_ctx *c, const int64_t *sk, const uint64_t *pt_rows, int level, uint64_t seed, uint64_t *ct) {
    int N = c->N; size_t n = (size_t)N;
    int64_t *e = malloc(sizeof(int64_t) * n); gauss(seed ^ 0xABCDEFull, N, e);
    uint64_t *s = malloc(sizeof(uint64_t) * n), *t = malloc(sizeof(uint64_t) * n);
    for (int l = 0; l <= level; l++) {
        const or_mod *m = &c->m[l];
        uint64_t *c0 = ct + (size_t)l * n, *c1 = ct + ((size_t)(level + 1) + (size_t)l) * n;
        or_fill_seeded(seed + 0x2000 + (uint64_t)l, m->q, N, c1);
        or_sk_rows(c, sk, l, s);
        signed_rows(c, e, l, t);
        for (int j = 0; j < N; j++) t[j] = addmod(t[j], pt_rows[(size_t)l * n + (size_t)j] % m->q, m->q);
        or_ntt(c, l, t, t);
        for (int j = 0; j < N; j++) c0[j] = submod(t[j], mulmod(c1[j], s[j], m->q), m->q);
    }
    free(e); free(s); free(t);
}
/* Decrypt (c0 + c1*s), InvNTT, centre mod Q0, divide by scale (ckks DecodeCoeffs at level 0) */
void or_decrypt_decode_l0(const or_ctx *c, const int64_t *sk, const uint64_t *ct, double scale, double *out) {
    int N = c->N; size_t n = (size_t)N; const or_mod *m = &c->m[0];
    uint64_t *s = malloc(sizeof(uint64_t) * n), *t = malloc(sizeof(uint64_t) * n);
    or_sk_rows(c, sk, 0, s);
    for (int j = 0; j < N; j++) t[j] = addmod(ct[j], mulmod(ct[n + (size_t)j], s[j], m->q), m->q);
    or_intt(c, 0, t, t);
    for (int j = 0; j < N; j++) {
        uint64_t v = t[j];
        double d = v > m->q / 2 ? -(double)(m->q - v) : (double)v;
        out[j] = d / scale;
    }
    free(s); free(t);
}
