/*
 * gotrace — ptrace harness that pins the oracle against the reference's own prebuilt binary.
 *
 * TEST INFRASTRUCTURE ONLY. Runs only in the build container (needs /root/reference/test_run);
 * its output (tests/golden/ref_trace_*.json: seeds + SHA-256 digests, no reference text) is what travels.
 *
 * Why: the reference (Go 1.16.6 + Lattigo fork test_lattigo@eb33b0555aaa) draws keys and encryption noise
 * from crypto-random, serialises nothing and has no tests, so nothing in its tree pins ciphertext
 * coefficients. The binary itself, however, runs here. This tool starts it under
 * ptrace, and at the entry of main.conv_then_pack (conv.go:522) OVERWRITES every input of the hot path —
 * ct_in (conv.go:527), pl_ker[i] (conv.go:527) and, at the first use of each Galois key inside
 * rlwe.(*KeySwitcher).SwitchKeysInPlace, the level-0 slices of that switching key (conv.go:291) — with
 * residues derived from a counter-based splitmix64 stream. It then records SHA-256 digests of
 *   - every plain_idx[s] (conv.go:241-261; pins Lattigo's psi choice + NTT ordering),
 *   - the ciphertext after each MulNew / SetScale (conv.go:527-528),
 *   - per pack-tree node: MulNew, SubNew, Add, SwitchKeysInPlace(p0,p1), RotateGal, Add (conv.go:288-292),
 *   - the ciphertext returned by conv_then_pack (conv.go:545) and its Scale.
 * The oracle replays the same seeds and must reproduce every digest (tests/test_oracle_pin.py).
 *
 * Go 1.16 uses the stack ABI0: at a function's first instruction [rsp] = return address and the
 * arguments (receiver first) start at [rsp+8]; results follow the arguments.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <inttypes.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ptrace.h>
#include <sys/types.h>
#include <sys/user.h>
#include <sys/wait.h>
#include <unistd.h>

/* ---------- sha256 (FIPS 180-4) ---------- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill; } sha256_t;
static const uint32_t K256[64] = {
0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,
0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,
0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,
0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x,n) (((x)>>(n))|((x)<<(32-(n))))
static void sha_block(sha256_t *s, const uint8_t *p) {
    uint32_t w[64], a,b,c,d,e,f,g,h;
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4*i]<<24 | (uint32_t)p[4*i+1]<<16 | (uint32_t)p[4*i+2]<<8 | p[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i-15],7)^ROR(w[i-15],18)^(w[i-15]>>3), s1 = ROR(w[i-2],17)^ROR(w[i-2],19)^(w[i-2]>>10);
        w[i] = w[i-16]+s0+w[i-7]+s1;
    }
    a=s->h[0];b=s->h[1];c=s->h[2];d=s->h[3];e=s->h[4];f=s->h[5];g=s->h[6];h=s->h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1=ROR(e,6)^ROR(e,11)^ROR(e,25), ch=(e&f)^(~e&g), t1=h+S1+ch+K256[i]+w[i];
        uint32_t S0=ROR(a,2)^ROR(a,13)^ROR(a,22), mj=(a&b)^(a&c)^(b&c), t2=S0+mj;
        h=g;g=f;f=e;e=d+t1;d=c;c=b;b=a;a=t1+t2;
    }
    s->h[0]+=a;s->h[1]+=b;s->h[2]+=c;s->h[3]+=d;s->h[4]+=e;s->h[5]+=f;s->h[6]+=g;s->h[7]+=h;
}
static void sha_init(sha256_t *s) {
    static const uint32_t iv[8]={0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    memcpy(s->h, iv, sizeof iv); s->len = 0; s->fill = 0;
}
static void sha_update(sha256_t *s, const void *data, size_t n) {
    const uint8_t *p = data; s->len += n;
    while (n) {
        size_t k = 64 - s->fill; if (k > n) k = n;
        memcpy(s->buf + s->fill, p, k); s->fill += k; p += k; n -= k;
        if (s->fill == 64) { sha_block(s, s->buf); s->fill = 0; }
    }
}
static void sha_final(sha256_t *s, char hex[65]) {
    uint64_t bits = s->len * 8; uint8_t pad = 0x80; sha_update(s, &pad, 1);
    uint8_t z = 0; while (s->fill != 56) sha_update(s, &z, 1);
    uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (56 - 8*i));
    sha_update(s, l, 8);
    for (int i = 0; i < 8; i++) sprintf(hex + 8*i, "%08x", s->h[i]);
}

/* ---------- counter-based splitmix64 stream (same function in tests/seedgen.py and oracle) ---------- */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* ---------- tracee memory ---------- */
static pid_t g_pid; static int g_mem = -1;
static void rd(uint64_t addr, void *buf, size_t n) {
    if (pread(g_mem, buf, n, (off_t)addr) != (ssize_t)n) { fprintf(stderr, "rd %#lx+%zu failed: %s\n", addr, n, strerror(errno)); exit(3); }
}
static void wr(uint64_t addr, const void *buf, size_t n) {
    if (pwrite(g_mem, buf, n, (off_t)addr) != (ssize_t)n) { fprintf(stderr, "wr %#lx+%zu failed: %s\n", addr, n, strerror(errno)); exit(3); }
}
static uint64_t rd64(uint64_t a) { uint64_t v; rd(a, &v, 8); return v; }
static double rdf64(uint64_t a) { double v; rd(a, &v, 8); return v; }

/* ---------- breakpoints ---------- */
typedef void (*handler_t)(pid_t tid, struct user_regs_struct *r, void *ud);
typedef struct { uint64_t addr; uint8_t orig; int armed; handler_t fn; void *ud; int refs; } bp_t;
#define MAXBP 256
static bp_t g_bp[MAXBP]; static int g_nbp;
static bp_t *bp_find(uint64_t a) { for (int i = 0; i < g_nbp; i++) if (g_bp[i].addr == a && g_bp[i].refs > 0) return &g_bp[i]; return NULL; }
static void bp_arm(bp_t *b) { uint8_t cc = 0xcc; rd(b->addr, &b->orig, 1); wr(b->addr, &cc, 1); b->armed = 1; }
static void bp_disarm(bp_t *b) { if (b->armed) { wr(b->addr, &b->orig, 1); b->armed = 0; } }
static bp_t *bp_add(uint64_t a, handler_t fn, void *ud) {
    bp_t *b = bp_find(a);
    if (b) { b->refs++; return b; }
    for (int i = 0; i < g_nbp; i++) if (g_bp[i].refs == 0) { b = &g_bp[i]; break; }
    if (!b) { if (g_nbp == MAXBP) { fprintf(stderr, "too many bps\n"); exit(3); } b = &g_bp[g_nbp++]; }
    b->addr = a; b->fn = fn; b->ud = ud; b->refs = 1; bp_arm(b); return b;
}
static void bp_release(bp_t *b) { if (--b->refs == 0) bp_disarm(b); }

/* pending function returns: LIFO per return address (Go may relocate the goroutine stack between
 * entry and return, so the stack pointer cannot be used to match them). */
typedef struct { uint64_t ret_addr; handler_t fn; void *ud; } pend_t;
#define MAXPEND 256
static pend_t g_pend[MAXPEND]; static int g_npend;
static void on_return_bp(pid_t tid, struct user_regs_struct *r, void *ud);
static void hook_return(struct user_regs_struct *r, handler_t fn, void *ud) {
    uint64_t ra = rd64(r->rsp);
    if (g_npend == MAXPEND) { fprintf(stderr, "pend overflow\n"); exit(3); }
    g_pend[g_npend++] = (pend_t){ra, fn, ud};
    bp_add(ra, on_return_bp, NULL);
}
static void on_return_bp(pid_t tid, struct user_regs_struct *r, void *ud) {
    (void)ud;
    uint64_t a = r->rip;
    for (int i = g_npend - 1; i >= 0; i--) if (g_pend[i].ret_addr == a) {
        pend_t p = g_pend[i];
        memmove(&g_pend[i], &g_pend[i+1], (size_t)(g_npend - i - 1) * sizeof(pend_t)); g_npend--;
        bp_t *b = bp_find(a); p.fn(tid, r, p.ud); if (b) bp_release(b);
        return;
    }
}

/* ---------- Go object walkers (layouts verified at run time by `probe`) ---------- */
/* ring.Poly{Coeffs [][]uint64, ...}: word0 = ptr to slice headers, word1 = len (limbs) */
static int poly_limbs(uint64_t poly) { return (int)rd64(poly + 8); }
static uint64_t poly_row(uint64_t poly, int limb, uint64_t *len) {
    uint64_t hdr = rd64(poly) + 24ull * (uint64_t)limb;
    if (len) *len = rd64(hdr + 8);
    return rd64(hdr);
}
/* ckks.Ciphertext{*rlwe.Ciphertext, Scale}; rlwe.Ciphertext{Value []*ring.Poly} */
static int ct_degree1(uint64_t ct) { return (int)rd64(rd64(ct) + 8); }
static uint64_t ct_poly(uint64_t ct, int k) { return rd64(rd64(rd64(ct)) + 8ull * (uint64_t)k); }
static double ct_scale(uint64_t ct) { return rdf64(ct + 8); }
/* ckks.Plaintext{*rlwe.Plaintext, Scale}; rlwe.Plaintext{Value *ring.Poly} */
static uint64_t pt_poly(uint64_t pt) { return rd64(rd64(pt)); }
static double pt_scale(uint64_t pt) { return rdf64(pt + 8); }

static FILE *g_out; static int g_first_event = 1;
static uint64_t g_N = 65536;
static uint64_t *g_tmp;

static void hash_rows(sha256_t *s, uint64_t poly, int nlimbs) {
    for (int l = 0; l < nlimbs; l++) {
        uint64_t len, p = poly_row(poly, l, &len);
        if (len != g_N) { fprintf(stderr, "row len %lu != N\n", len); exit(3); }
        rd(p, g_tmp, g_N * 8); sha_update(s, g_tmp, g_N * 8);
    }
}
static void emit_begin(const char *op) {
    fprintf(g_out, "%s\n  {\"op\": \"%s\"", g_first_event ? "" : ",", op); g_first_event = 0;
}
static void emit_poly(const char *key, uint64_t poly, int nlimbs) {
    sha256_t s; char hex[65]; sha_init(&s); hash_rows(&s, poly, nlimbs); sha_final(&s, hex);
    uint64_t first[4]; rd(poly_row(poly, 0, NULL), first, 32);
    fprintf(g_out, ", \"%s\": {\"limbs\": %d, \"sha256\": \"%s\", \"head\": [%lu, %lu, %lu, %lu]}", key, nlimbs, hex,
            first[0], first[1], first[2], first[3]);
}
static void emit_ct(const char *key, uint64_t ct) {
    int deg1 = ct_degree1(ct); int limbs = poly_limbs(ct_poly(ct, 0));
    fprintf(g_out, ", \"%s\": {\"scale\": %.17g, \"level\": %d, \"polys\": [", key, ct_scale(ct), limbs - 1);
    for (int k = 0; k < deg1; k++) {
        sha256_t s; char hex[65]; sha_init(&s); hash_rows(&s, ct_poly(ct, k), limbs); sha_final(&s, hex);
        uint64_t first[2]; rd(poly_row(ct_poly(ct, k), 0, NULL), first, 16);
        fprintf(g_out, "%s{\"sha256\": \"%s\", \"head\": [%lu, %lu]}", k ? ", " : "", hex, first[0], first[1]);
    }
    fprintf(g_out, "]}");
}
static void emit_end(void) { fprintf(g_out, "}"); fflush(g_out); }

static int g_noplant = 0;   /* -noplant: log levels / scales / digests of the run's own data, overwrite nothing */
static void plant_row(uint64_t rowptr, uint64_t seed, uint64_t q) {
    if (g_noplant) return;
    for (uint64_t j = 0; j < g_N; j++) g_tmp[j] = splitmix64_at(seed, j) % q;
    wr(rowptr, g_tmp, g_N * 8);
}

/* ---------- addresses (nm /root/reference/test_run); each hook sits on the first instruction AFTER the
 * goroutine stack-growth check (the `sub $frame,%rsp`), where rsp still equals the entry rsp and the
 * function cannot be restarted by runtime.morestack any more ---------- */
#define A_CALL_BL            0x548544ull  /* call main.testConv_BL_in in main.main (main.go:640) */
#define A_CONV_THEN_PACK     0x53b17bull
#define A_MULNEW             0x5227dbull
#define A_SETSCALE           0x521b98ull
#define A_SUBNEW             0x51b9b3ull
#define A_ADD                0x51aef3ull
#define A_ROTATEGAL          0x5245b3ull
#define A_SWITCHKEYS         0x4fdd53ull
#define A_MULTBYCONST        0x51ee98ull
#define A_DIVROUND           0x4f3ab3ull
#define A_ENCODECOEFFS       0x518bb8ull
#define A_RESCALE            0x522453ull  /* ckks.(*evaluator).Rescale */
#define A_MULRELIN           0x522c7bull  /* ckks.(*evaluator).mulRelin (behind Mul / MulRelin / MulNew / MulRelinNew) */
#define A_ROTATE             0x524438ull  /* ckks.(*evaluator).Rotate(ct0, k, ctOut) */
#define A_MODUP              0x50741bull  /* ckks.(*Bootstrapper).modUp(ct) *Ciphertext */
#define A_INVFFT             0x5185d3ull  /* ckks.invfft(values []complex128, N, M uint64, rotGroup []uint64, roots []complex128) */
#define A_ENCODE             0x514cf3ull  /* ckks.(*encoderComplex128).Encode(pt *Plaintext, values []complex128, logSlots uint64) */
#define A_EVALPOLY           0x52d3dbull  /* ckks.(*evaluator).EvaluatePoly(ct0 *Ciphertext, pol *Poly, targetScale float64) (*Ciphertext, error) */
#define A_MGIAA              0x520060ull  /* ckks.(*evaluator).MultByGaussianIntegerAndAdd(ct0, cReal, cImag int64, ctOut) */
#define A_ADDCONST           0x51d9f3ull  /* ckks.(*evaluator).AddConst(ct0, constant interface{}, ctOut) */
#define A_DROPLEVEL          0x5223a0ull  /* ckks.(*evaluator).DropLevel(ct0, levels uint64) */
#define A_RECURSE            0x52eb18ull  /* ckks.recurse(targetScale, logSplit, logDegree, coeffs *Poly, C, evaluator) */
#define A_POLYLEAF           0x52fd1bull  /* ckks.evaluatePolyFromPowerBasis(targetScale, coeffs *Poly, C, evaluator) */
#define A_ENCDIAG            0x517ab8ull  /* ckks.(*encoderComplex128).encodeDiagonal(logSlots, level, scale, m []complex128) [2]*ring.Poly */
#define A_ENCMAT             0x5173bbull  /* ckks.(*encoderComplex128).EncodeDiagMatrixBSGSAtLvl(level, diagMatrix, scale, maxN1N2Ratio, logSlots) *PtDiagMatrix */
#define A_POWERBASIS         0x52dbd3ull  /* ckks.computePowerBasis(n, C, scale, evaluator) */
#define A_TYPE_FLOAT64       0x570a20ull  /* runtime type descriptor of float64 (seen in the interface word) */

static const uint64_t Q0 = 0x80000000080001ull, Q1 = 0x1ffffffea0001ull, P0 = 0x1fffffffffe00001ull;
static uint64_t g_seed = 0xC0FFEE;
static int g_mode_probe, g_in_ctp, g_verbose, g_lean, g_after_ctp, g_encode_calls;
static int g_skip_bl = 1;
/* general key-switch trace mode (-ks N -Q q0,q1,.. -P p0,..): plants and records the first N SwitchKeysInPlace calls at ANY level */
static int g_ks_max = 0, g_ks_calls = 0, g_nQ = 0, g_nP = 0, g_nQ_total = 0, g_nQ_full = 0, g_ks_unique = 0; static uint64_t g_Q[64], g_Pm[16];
#define SEED_KSX(call,l)        (g_seed + ((4ull<<32) | (uint64_t)((call)*64+(l))))
#define SEED_KSEVK(id,d,k,li)   (g_seed + ((5ull<<32) | (uint64_t)((((id)*32+(d))*2+(k))*64+(li))))

/* seed lanes: tag<<32 | index */
#define SEED_CT(p,l)      (g_seed + ((1ull<<32) | (uint64_t)((p)*8+(l))))
#define SEED_KER(i,l)     (g_seed + ((2ull<<32) | (uint64_t)((i)*8+(l))))
#define SEED_EVK(k,c,w)   (g_seed + ((3ull<<32) | (uint64_t)((k)*8+(c)*2+(w))))

static void dump_words(const char *what, uint64_t a, int n) {
    fprintf(stderr, "%s @%#lx:", what, a);
    for (int i = 0; i < n; i++) fprintf(stderr, " %#lx", rd64(a + 8ull*(uint64_t)i));
    fprintf(stderr, "\n");
}

/* --- evaluator-op hooks, active only inside conv_then_pack --- */
typedef struct { uint64_t a, b, c; } ud3_t;
static ud3_t g_udpool[MAXPEND]; static int g_udi;
static ud3_t *ud_new(uint64_t a, uint64_t b, uint64_t c) { ud3_t *u = &g_udpool[g_udi++ % MAXPEND]; u->a = a; u->b = b; u->c = c; return u; }

static void ret_mulnew(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t ct = rd64(r->rsp - 8 + 0x30); emit_begin("MulNew"); emit_ct("out", ct); emit_end(); }
static void on_mulnew(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (g_in_ctp && !g_lean) hook_return(r, ret_mulnew, NULL); }

static void ret_setscale(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    emit_begin("SetScale"); emit_ct("out", ((ud3_t*)ud)->a); emit_end(); }
static void on_setscale(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_ctp) return;
    hook_return(r, ret_setscale, ud_new(rd64(r->rsp + 0x10), 0, 0)); }

static void ret_subnew(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t ct = rd64(r->rsp - 8 + 0x30); emit_begin("SubNew"); emit_ct("out", ct); emit_end(); }
static void on_subnew(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (g_in_ctp && !g_lean) hook_return(r, ret_subnew, NULL); }

static void ret_add(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    emit_begin(((ud3_t*)ud)->b ? "Add.bias" : "Add"); emit_ct("out", ((ud3_t*)ud)->a); emit_end(); }
static int g_add_calls;
static int g_in_poly_fwd(void);
static void on_p_add(pid_t t, struct user_regs_struct *r, void *ud);
static void on_p_multbyconst(pid_t t, struct user_regs_struct *r, void *ud);
static void on_add(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (g_in_poly_fwd()) { on_p_add(t, r, ud); return; }
    /* Add(recv, op0 (itab,ptr), op1 (itab,ptr), ctOut) */
    if (g_after_ctp) {           /* eval.go:258  Add(ct_res, pl_bn_b, ct_res) */
        g_after_ctp = 0;
        uint64_t pt = rd64(r->rsp + 0x28);
        emit_begin("bias_plaintext"); fprintf(g_out, ", \"scale\": %.17g", pt_scale(pt));
        emit_poly("pt", pt_poly(pt), poly_limbs(pt_poly(pt))); emit_end();
        hook_return(r, ret_add, ud_new(rd64(r->rsp + 0x30), 1, 0)); return;
    }
    if (!g_in_ctp) return;
    int nth = g_add_calls++;     /* conv.go:290 (even) and conv.go:292 (odd, the node result) */
    if (g_lean && !(nth & 1)) return;
    hook_return(r, ret_add, ud_new(rd64(r->rsp + 0x30), 0, 0)); }

static void ret_rotgal(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    emit_begin("RotateGal"); fprintf(g_out, ", \"galEl\": %lu", ((ud3_t*)ud)->b); emit_ct("out", ((ud3_t*)ud)->a); emit_end(); }
static void on_rotgal(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_ctp || g_lean) return;
    hook_return(r, ret_rotgal, ud_new(rd64(r->rsp + 0x20), rd64(r->rsp + 0x18), 0)); }

static void ret_multbyconst(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ud3_t *u = ud; double c; memcpy(&c, &u->b, 8);
    emit_begin("MultByConst"); fprintf(g_out, ", \"const_is_f64\": %d, \"const\": %.17g", (int)u->c, c);
    emit_ct("out", u->a); emit_end(); }
static void on_multbyconst(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (g_in_poly_fwd()) { on_p_multbyconst(t, r, ud); return; }
    if (!g_in_ctp || g_lean) return;
    /* MultByConst(recv, ct0 *Ciphertext, constant interface{} (2 words), ctOut *Ciphertext) */
    if (g_mode_probe) dump_words("MultByConst args", r->rsp + 8, 6);
    uint64_t ty = rd64(r->rsp + 0x18), data = rd64(r->rsp + 0x20);
    int is_f64 = (ty == A_TYPE_FLOAT64);
    hook_return(r, ret_multbyconst, ud_new(rd64(r->rsp + 0x28), is_f64 ? rd64(data) : 0, (uint64_t)is_f64)); }

/* SwitchKeysInPlace(recv, levelQ int, cx *ring.Poly, evakey *rlwe.SwitchingKey, p0, p1 *ring.Poly) */
static uint64_t g_evk_seen[64]; static int g_nevk;
static void ret_switchkeys(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ud3_t *u = ud; emit_begin("SwitchKeysInPlace"); fprintf(g_out, ", \"evk\": %lu", u->c);
    emit_poly("p0", u->a, 1); emit_poly("p1", u->b, 1); emit_end(); }
typedef struct { uint64_t p0, p1; int level, evk, call, alpha, beta; } ksrec_t;
static ksrec_t g_ksrec[8]; static int g_ksrec_i;
static void ret_switchkeys_general(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ksrec_t *u = ud; emit_begin("SwitchKeysInPlace.general");
    fprintf(g_out, ", \"call\": %d, \"level\": %d, \"evk\": %d, \"alpha\": %d, \"beta\": %d", u->call, u->level, u->evk, u->alpha, u->beta);
    emit_poly("p0", u->p0, u->level + 1); emit_poly("p1", u->p1, u->level + 1); emit_end();
    if (g_ks_calls >= g_ks_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); }
}
static void on_switchkeys_general(struct user_regs_struct *r) {
    uint64_t level = rd64(r->rsp + 0x10), cx = rd64(r->rsp + 0x18), evk = rd64(r->rsp + 0x20);
    uint64_t p0 = rd64(r->rsp + 0x28), p1 = rd64(r->rsp + 0x30);
    if (g_ks_calls >= g_ks_max || (int)level >= g_nQ) return;
    /* alpha = number of special-prime limbs of THIS key (the run may hold evaluators with different P) */
    int alpha = g_nP;
    { uint64_t v0 = rd64(evk); int limbs0 = poly_limbs(rd64(v0)); if (g_nQ_full) alpha = limbs0 - g_nQ_full; }
    if (alpha < 1 || alpha > g_nP) return;
    { static int seen[64][8]; if (g_ks_unique && seen[level][alpha] >= g_ks_unique) return; seen[level][alpha]++; }
    int id = -1; for (int i = 0; i < g_nevk; i++) if (g_evk_seen[i] == evk) id = i;
    if (id < 0) { if (g_nevk >= 64) return; id = g_nevk; g_evk_seen[g_nevk++] = evk; }
    const int beta = ((int)level + 1 + alpha - 1) / alpha, call = g_ks_calls++;
    if (poly_limbs(cx) < (int)level + 1) { fprintf(stderr, "cx has %d limbs < level+1\n", poly_limbs(cx)); exit(3); }
    for (int l = 0; l <= (int)level; l++) plant_row(poly_row(cx, l, NULL), SEED_KSX(call, l), g_Q[l]);
    uint64_t v = rd64(evk);                                   /* Value [][2]*ring.Poly */
    for (int d = 0; d < beta; d++) for (int k = 0; k < 2; k++) {
        uint64_t poly = rd64(v + 16ull * (uint64_t)d + 8ull * (uint64_t)k); int limbs = poly_limbs(poly);
        if (g_nQ_total == 0) g_nQ_total = limbs - g_nP;
        for (int l = 0; l <= (int)level; l++) plant_row(poly_row(poly, l, NULL), SEED_KSEVK(id, d, k, l), g_Q[l]);
        for (int j = 0; j < alpha; j++) plant_row(poly_row(poly, limbs - alpha + j, NULL), SEED_KSEVK(id, d, k, 32 + j), g_Pm[j]);
    }
    ksrec_t *u = &g_ksrec[g_ksrec_i++ % 8]; u->p0 = p0; u->p1 = p1; u->level = (int)level; u->evk = id; u->call = call; u->alpha = alpha; u->beta = beta;
    fprintf(stderr, "KS call %d level %lu evk %d beta %d\n", call, level, id, beta);
    hook_return(r, ret_switchkeys_general, u);
}
static void nested_ks_plant(struct user_regs_struct *r);
static int g_nested_ks_fwd(void);
static void on_switchkeys(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (g_nested_ks_fwd()) { nested_ks_plant(r); return; }
    if (g_ks_max) { on_switchkeys_general(r); return; }
    if (!g_in_ctp) return;
    uint64_t level = rd64(r->rsp + 0x10), cx = rd64(r->rsp + 0x18), evk = rd64(r->rsp + 0x20);
    uint64_t p0 = rd64(r->rsp + 0x28), p1 = rd64(r->rsp + 0x30);
    int k = -1; for (int i = 0; i < g_nevk; i++) if (g_evk_seen[i] == evk) k = i;
    if (g_mode_probe) {
        fprintf(stderr, "SwitchKeysInPlace level=%lu cx=%#lx evk=%#lx p0=%#lx p1=%#lx\n", level, cx, evk, p0, p1);
        dump_words(" evk", evk, 4); uint64_t v = rd64(evk); dump_words(" evk.Value[0]", v, 4);
        dump_words("  poly b", rd64(v), 6); fprintf(stderr, "  limbs b=%d p0 limbs=%d cx limbs=%d\n", poly_limbs(rd64(v)), poly_limbs(p0), poly_limbs(cx));
    }
    if (level != 0) { fprintf(stderr, "unexpected level %lu in SwitchKeysInPlace\n", level); exit(3); }
    if (k < 0) {
        k = g_nevk; g_evk_seen[g_nevk++] = evk;
        uint64_t digits = rd64(evk + 8), v = rd64(evk);               /* Value [][2]*ring.Poly */
        (void)digits;
        for (int c = 0; c < 2; c++) {
            uint64_t poly = rd64(v + 8ull*(uint64_t)c); int limbs = poly_limbs(poly);
            plant_row(poly_row(poly, 0, NULL), SEED_EVK(k, c, 0), Q0);
            plant_row(poly_row(poly, limbs - 1, NULL), SEED_EVK(k, c, 1), P0);   /* single P prime = last limb */
        }
    }
    if (!g_lean) hook_return(r, ret_switchkeys, ud_new(p0, p1, (uint64_t)k));
}

/* ---------- -ops N: the leveled evaluator of the convReLU chain. Plants both inputs of ckks.(*evaluator).mulRelin (relin = true,
 * two degree-1 ciphertexts) and the input of ckks.(*evaluator).Rescale, first `-ops-unique` calls per (operation, level), and
 * records level / scale / SHA-256 of the result. The relinearisation key the nested SwitchKeysInPlace reads is planted with the
 * same seeds the -ks mode uses (SEED_KSEVK) and its id is recorded, the key-switch input is left alone (it is the real c2). */
#define SEED_OPIN(call, operand, poly, limb) (g_seed + ((6ull << 32) | (((((uint64_t)(call)) * 2 + (uint64_t)(operand)) * 4 + (uint64_t)(poly)) * 64 + (uint64_t)(limb))))
static int g_ops_max = 0, g_ops_calls = 0, g_ops_unique = 1, g_nested_ks = 0, g_nested_evk = -1, g_nested_alpha = 0;
static int g_nested_ks_fwd(void) { return g_nested_ks; }
typedef struct { uint64_t out; int call, level; double s0, s1, min_scale; } oprec_t;
static oprec_t g_oprec[8]; static int g_oprec_i;
static void ops_done_check(void) { if (g_ops_calls >= g_ops_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); } }
static void plant_ct(uint64_t ct, int call, int operand) {
    int limbs = poly_limbs(ct_poly(ct, 0));
    for (int k = 0; k < 2; k++) for (int l = 0; l < limbs; l++) plant_row(poly_row(ct_poly(ct, k), l, NULL), SEED_OPIN(call, operand, k, l), g_Q[l]);
}
static void nested_ks_plant(struct user_regs_struct *r) {      /* inside a traced mulRelin: plant the key rows only */
    uint64_t level = rd64(r->rsp + 0x10), evk = rd64(r->rsp + 0x20);
    int alpha = g_nP; { uint64_t v0 = rd64(evk); int limbs0 = poly_limbs(rd64(v0)); if (g_nQ_full) alpha = limbs0 - g_nQ_full; }
    int id = -1; for (int i = 0; i < g_nevk; i++) if (g_evk_seen[i] == evk) id = i;
    if (id < 0) { id = g_nevk; g_evk_seen[g_nevk++] = evk; }
    const int beta = ((int)level + 1 + alpha - 1) / alpha; uint64_t v = rd64(evk);
    for (int d = 0; d < beta; d++) for (int k = 0; k < 2; k++) {
        uint64_t poly = rd64(v + 16ull * (uint64_t)d + 8ull * (uint64_t)k); int limbs = poly_limbs(poly);
        for (int l = 0; l <= (int)level; l++) plant_row(poly_row(poly, l, NULL), SEED_KSEVK(id, d, k, l), g_Q[l]);
        for (int j = 0; j < alpha; j++) plant_row(poly_row(poly, limbs - alpha + j, NULL), SEED_KSEVK(id, d, k, 32 + j), g_Pm[j]);
    }
    g_nested_evk = id; g_nested_alpha = alpha;
}
static void ret_mulrelin(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    oprec_t *u = ud; g_nested_ks = 0;
    emit_begin("MulRelin"); fprintf(g_out, ", \"call\": %d, \"level\": %d, \"square\": %d, \"scale0\": %.17g, \"scale1\": %.17g, \"evk\": %d, \"alpha\": %d", u->call, u->level, (int)u->min_scale, u->s0, u->s1, g_nested_evk, g_nested_alpha);
    emit_ct("out", u->out); emit_end(); ops_done_check(); }
static void on_mulrelin(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_ops_max || g_nested_ks) return;
    /* mulRelin(recv, op0 Operand (itab, ptr), op1 Operand (itab, ptr), relin bool, ctOut *Ciphertext) */
    uint64_t op0 = rd64(r->rsp + 0x18), op1 = rd64(r->rsp + 0x28), out = rd64(r->rsp + 0x38); uint8_t relin; rd(r->rsp + 0x30, &relin, 1);
    if (!relin || ct_degree1(op0) != 2 || ct_degree1(op1) != 2) return;
    int l0 = poly_limbs(ct_poly(op0, 0)) - 1, l1 = poly_limbs(ct_poly(op1, 0)) - 1, level = l0 < l1 ? l0 : l1;
    if (l0 != l1 || level >= g_nQ) return;                      /* equal levels only: keeps the replay unambiguous */
    { static int seen[64]; if (seen[level] >= g_ops_unique) return; seen[level]++; }
    int call = g_ops_calls++;
    plant_ct(op0, call, 0); if (op1 != op0) plant_ct(op1, call, 1);
    oprec_t *u = &g_oprec[g_oprec_i++ % 8]; u->out = out; u->call = call; u->level = level; u->s0 = ct_scale(op0); u->s1 = ct_scale(op1); u->min_scale = op1 == op0 ? 1 : 0;
    g_nested_ks = 1; g_nested_evk = -1;
    fprintf(stderr, "MulRelin call %d level %d%s\n", call, level, op1 == op0 ? " (square)" : "");
    hook_return(r, ret_mulrelin, u);
}
static void ret_rescale(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    oprec_t *u = ud;
    emit_begin("Rescale"); fprintf(g_out, ", \"call\": %d, \"level\": %d, \"scale_in\": %.17g, \"min_scale\": %.17g", u->call, u->level, u->s0, u->min_scale);
    emit_ct("out", u->out); emit_end(); ops_done_check(); }
static void on_rescale(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_ops_max || g_nested_ks) return;
    /* Rescale(recv, ctIn *Ciphertext, minScale float64, ctOut *Ciphertext) */
    uint64_t in = rd64(r->rsp + 0x10), out = rd64(r->rsp + 0x20); double ms = rdf64(r->rsp + 0x18);
    if (ct_degree1(in) != 2) return;
    int level = poly_limbs(ct_poly(in, 0)) - 1;
    if (level < 1 || level >= g_nQ) return;
    { static int seen[64]; if (seen[level] >= g_ops_unique) return; seen[level]++; }
    int call = g_ops_calls++;
    plant_ct(in, call, 0);
    oprec_t *u = &g_oprec[g_oprec_i++ % 8]; u->out = out; u->call = call; u->level = level; u->s0 = ct_scale(in); u->min_scale = ms;
    fprintf(stderr, "Rescale call %d level %d scale %g\n", call, level, u->s0);
    hook_return(r, ret_rescale, u);
}

/* Rotate(recv, ct0 *Ciphertext, k int, ctOut *Ciphertext): planted input, planted rotation key (nested), first call per level */
static void ret_rotate(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    oprec_t *u = ud; g_nested_ks = 0;
    emit_begin("Rotate"); fprintf(g_out, ", \"call\": %d, \"level\": %d, \"k\": %ld, \"evk\": %d, \"alpha\": %d", u->call, u->level, (long)u->s0, g_nested_evk, g_nested_alpha);
    emit_ct("out", u->out); emit_end(); ops_done_check(); }
static void on_rotate(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_ops_max || g_nested_ks) return;
    uint64_t in = rd64(r->rsp + 0x10), out = rd64(r->rsp + 0x20); int64_t k = (int64_t)rd64(r->rsp + 0x18);
    if (ct_degree1(in) != 2) return;
    int level = poly_limbs(ct_poly(in, 0)) - 1;
    if (level >= g_nQ) return;
    { static int seen[64]; if (seen[level] >= g_ops_unique) return; seen[level]++; }
    int call = g_ops_calls++;
    plant_ct(in, call, 0);
    oprec_t *u = &g_oprec[g_oprec_i++ % 8]; u->out = out; u->call = call; u->level = level; u->s0 = (double)k;
    g_nested_ks = 1; g_nested_evk = -1;
    fprintf(stderr, "Rotate call %d level %d k %ld\n", call, level, (long)k);
    hook_return(r, ret_rotate, u);
}
/* modUp(recv *Bootstrapper, ct *Ciphertext) *Ciphertext: planted level-0 input, result read from the return slot */
static void ret_modup(pid_t t, struct user_regs_struct *r, void *ud) { (void)t;
    oprec_t *u = ud; uint64_t out = rd64(r->rsp - 8 + 0x18);
    emit_begin("modUp"); fprintf(g_out, ", \"call\": %d, \"level\": %d", u->call, u->level); emit_ct("out", out); emit_end(); ops_done_check(); }
static void on_modup(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_ops_max) return;
    { static int seen; if (seen >= g_ops_unique) return; seen++; }
    uint64_t in = rd64(r->rsp + 0x10); int level = poly_limbs(ct_poly(in, 0)) - 1, call = g_ops_calls++;
    plant_ct(in, call, 0);
    oprec_t *u = &g_oprec[g_oprec_i++ % 8]; u->call = call; u->level = level;
    fprintf(stderr, "modUp call %d input level %d\n", call, level);
    hook_return(r, ret_modup, u);
}

/* (*encoderComplex128).EncodeCoeffs(coeffs []float64, pt *ckks.Plaintext): digest of the encoded plaintext
 * (coefficient domain, before ToNTT). Call order in `conv k i n` with BL skipped: 16 x gen_idxNlogs
 * (conv.go:251), 1 x input (test.go:46), B x prep_Ker (conv.go:513), 1 x bias (eval.go:242). */
static void ret_encode(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ud3_t *u = ud; uint64_t pt = u->a;
    emit_begin("EncodeCoeffs"); fprintf(g_out, ", \"call\": %lu, \"ncoeffs\": %lu, \"scale\": %.17g", u->b, u->c, pt_scale(pt));
    emit_poly("pt", pt_poly(pt), poly_limbs(pt_poly(pt))); emit_end(); }
static void on_encode(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (g_ks_max) return;
    if (g_lean && g_encode_calls >= 24) { g_encode_calls++; return; }
    hook_return(r, ret_encode, ud_new(rd64(r->rsp + 0x28), (uint64_t)g_encode_calls++, rd64(r->rsp + 0x18))); }

/* -enc N: the slot encoder of the BL baseline run (needs -keep-bl): the first N ckks.invfft calls -- SHA-256 of the complex128
 * input and output vectors, and once of the encoder's root table (math.Cos / math.Sin of Go's runtime) and rotation group -- and
 * the plaintext every Encode call leaves (coefficient domain, scaleUpVecExact applied). Nothing is planted: the data are the
 * run's own (the CSVs tests/golden/gen_conv_csv.py writes), which the oracle regenerates. */
static int g_enc_max = 0, g_enc_calls = 0, g_encode_slots_calls = 0;
static void sha_mem(uint64_t addr, uint64_t bytes, char hex[65]) {
    sha256_t s; sha_init(&s); uint8_t buf[65536];
    while (bytes) { size_t k = bytes > sizeof buf ? sizeof buf : (size_t)bytes; rd(addr, buf, k); sha_update(&s, buf, k); addr += k; bytes -= k; }
    sha_final(&s, hex);
}
static void enc_done_check(void) { if (g_enc_calls >= g_enc_max && g_encode_slots_calls >= g_enc_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); } }
static void ret_invfft(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ud3_t *u = ud; char hex[65]; sha_mem(u->a, u->b * 16, hex);
    double head[4]; rd(u->a, head, 32);
    fprintf(g_out, ", \"out\": \"%s\", \"out_head\": [%.17g, %.17g, %.17g, %.17g]", hex, head[0], head[1], head[2], head[3]); emit_end(); enc_done_check(); }
static void on_invfft(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_enc_max || g_enc_calls >= g_enc_max) return;
    uint64_t vp = rd64(r->rsp + 8), vl = rd64(r->rsp + 0x10), Nn = rd64(r->rsp + 0x20), M = rd64(r->rsp + 0x28);
    uint64_t gp = rd64(r->rsp + 0x30), gl = rd64(r->rsp + 0x38), rp = rd64(r->rsp + 0x48), rl = rd64(r->rsp + 0x50);
    char hex[65];
    if (g_enc_calls == 0) {
        emit_begin("encoder_tables"); fprintf(g_out, ", \"N\": %lu, \"M\": %lu, \"roots_len\": %lu, \"rotgroup_len\": %lu", Nn, M, rl, gl);
        sha_mem(rp, rl * 16, hex); fprintf(g_out, ", \"roots\": \"%s\"", hex);
        sha_mem(rp, (rl - 1) * 16, hex); fprintf(g_out, ", \"roots_without_last\": \"%s\"", hex);
        sha_mem(gp, gl * 8, hex); fprintf(g_out, ", \"rotgroup\": \"%s\"", hex);
        double w[4]; rd(rp + 16, w, 32); fprintf(g_out, ", \"roots_1_2\": [%.17g, %.17g, %.17g, %.17g]", w[0], w[1], w[2], w[3]); emit_end();
    }
    emit_begin("invfft"); sha_mem(vp, vl * 16, hex);
    fprintf(g_out, ", \"call\": %d, \"n\": %lu, \"in\": \"%s\"", g_enc_calls++, vl, hex);
    hook_return(r, ret_invfft, ud_new(vp, vl, 0)); }
static void ret_encode_slots(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r;
    ud3_t *u = ud; uint64_t pt = u->a;
    emit_begin("Encode"); fprintf(g_out, ", \"call\": %lu, \"nvalues\": %lu, \"scale\": %.17g", u->b, u->c, pt_scale(pt));
    emit_poly("pt", pt_poly(pt), poly_limbs(pt_poly(pt))); emit_end(); enc_done_check(); }
static void on_encode_slots(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_enc_max || g_encode_slots_calls >= g_enc_max) return;
    hook_return(r, ret_encode_slots, ud_new(rd64(r->rsp + 0x10), (uint64_t)g_encode_slots_calls++, rd64(r->rsp + 0x20))); }

/* -diag N: the plaintext diagonals of the bootstrapper's DFT matrices (ckks.(*Bootstrapper).genDFTMatrices -> GenCoeffsToSlotsMatrix /
 * GenSlotsToCoeffsMatrix -> EncodeDiagMatrixBSGSAtLvl -> encodeDiagonal). Nothing is planted. Recorded per matrix: level, scale,
 * maxN1N2Ratio, logSlots and - on return - N1; per diagonal (Go iterates the index MAP, so the order inside a matrix is arbitrary):
 * level, scale, the SHA-256 of the complex128 values handed to the encoder and of the two encoded polynomials (mod Q: level+1 limbs in
 * NTT + Montgomery form; mod P). The run is killed after N diagonals or when the first convolution starts. -dump FILE additionally
 * writes the raw values (build-container analysis only; never committed). */
static int g_diag_max = 0, g_diag_calls = 0, g_diag_mats = 0; static FILE *g_dump;
static void diag_done(void) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); if (g_dump) fclose(g_dump); kill(g_pid, SIGKILL); exit(0); }
static void ret_encdiag(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; ud3_t *u = ud; (void)u;
    /* results sit above the arguments: [rsp-8+0x40], [rsp-8+0x48] relative to the entry rsp; here rsp = entry rsp + 8 */
    uint64_t mq = rd64(r->rsp + 0x38), mp = rd64(r->rsp + 0x40);
    emit_poly("mQ", mq, poly_limbs(mq)); emit_poly("mP", mp, poly_limbs(mp)); emit_end();
    if (g_diag_calls >= g_diag_max) diag_done(); }
static void on_encdiag(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_diag_max) return;
    uint64_t logslots = rd64(r->rsp + 0x10), level = rd64(r->rsp + 0x18), vp = rd64(r->rsp + 0x28), vl = rd64(r->rsp + 0x30); double scale = rdf64(r->rsp + 0x20);
    char hex[65]; sha_mem(vp, vl * 16, hex);
    emit_begin("encodeDiagonal"); fprintf(g_out, ", \"call\": %d, \"matrix\": %d, \"logSlots\": %lu, \"level\": %lu, \"scale\": %.17g, \"n\": %lu, \"values\": \"%s\"", g_diag_calls++, g_diag_mats - 1, logslots, level, scale, vl, hex);
    if (g_dump) { uint64_t hdr[4] = {(uint64_t)(g_diag_mats - 1), level, vl, 0}; memcpy(&hdr[3], &scale, 8); fwrite(hdr, 8, 4, g_dump);
                  uint8_t *buf = malloc(vl * 16); rd(vp, buf, vl * 16); fwrite(buf, 16, vl, g_dump); free(buf); fflush(g_dump); }
    hook_return(r, ret_encdiag, ud_new(0, 0, 0)); }
static void ret_encmat(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; ud3_t *u = ud;
    uint64_t m = rd64(r->rsp + 0x30);
    emit_begin("matrix_done"); fprintf(g_out, ", \"matrix\": %lu, \"LogSlots\": %lu, \"N1\": %lu, \"Level\": %lu, \"Scale\": %.17g", u->a, rd64(m), rd64(m + 8), rd64(m + 16), rdf64(m + 24)); emit_end(); }
static void on_encmat(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_diag_max) return;
    emit_begin("EncodeDiagMatrixBSGSAtLvl"); fprintf(g_out, ", \"matrix\": %d, \"level\": %lu, \"scale\": %.17g, \"maxN1N2Ratio\": %.17g, \"logSlots\": %lu", g_diag_mats, rd64(r->rsp + 0x10), rdf64(r->rsp + 0x20), rdf64(r->rsp + 0x28), rd64(r->rsp + 0x30)); emit_end();
    hook_return(r, ret_encmat, ud_new((uint64_t)g_diag_mats++, 0, 0)); }
static void on_ctp_diag(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r; (void)ud; if (g_diag_max) diag_done(); }
/* -logslots K (with -diag): make the binary build a SPARSE-slot bootstrapper, the one main.go:480-500 creates as btp2..btp5 for the resnet
 * (this snapshot of the binary never does). At the entry of ckks.NewBootstrapper_mod(params Parameters (by value, 13 words: logN at +0,
 * logSlots at +0x58 (newBootstrapper @5105c0 reads it there for dslots), btpParams *BootstrappingParameters, key) the two LogSlots
 * (the Parameters copy on the stack and btpParams+0xf8) are overwritten with K, which is what `params2 = ...; btpParams.LogSlots = LogN - 2`
 * amounts to. genDFTMatrices then runs on K and -diag records its diagonals; CheckKeys fails afterwards (the run's rotation keys are the
 * full-slot ones) and the run panics, which ends the trace. */
static int g_logslots = 0;
/* the same K for the rotation-key list: (*BootstrappingParameters).RotationsForBootstrapping(logSlots) is called by main.newContext
 * (main.go:466) before the keys are generated; with its argument patched the run owns the rotation keys of the sparse bootstrapper, CheckKeys
 * passes and a `convReLU` run goes on to call BootstrappConv_CtoS on it (-flow / -chain over the sparse bootstrapper) */
static void on_rots_for_btp(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_logslots) return;
    uint64_t k = (uint64_t)g_logslots, old = rd64(r->rsp + 0x10);
    wr(r->rsp + 0x10, &k, 8);
    emit_begin("RotationsForBootstrapping.patched"); fprintf(g_out, ", \"logSlots_was\": %lu, \"logSlots\": %lu", old, k); emit_end(); }
static void on_newbtp_mod(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_logslots) return;
    uint64_t k = (uint64_t)g_logslots, bp = rd64(r->rsp + 0x70);
    uint64_t old_p = rd64(r->rsp + 0x60), old_b = rd64(bp + 0xf8), logn = rd64(r->rsp + 0x8);
    wr(r->rsp + 0x60, &k, 8); wr(bp + 0xf8, &k, 8);
    emit_begin("NewBootstrapper_mod.patched"); fprintf(g_out, ", \"logN\": %lu, \"params_logSlots_was\": %lu, \"btpParams_LogSlots_was\": %lu, \"LogSlots\": %lu", logn, old_p, old_b, k); emit_end(); }

/* -flow: log-only walk through the convReLU chain between the entry of ckks.(*Bootstrapper).BootstrappConv_CtoS (eval.go:450) and the
 * return of main.evalConv_BNRelu_new (eval.go:272-607; BootstrappConv_StoC, eval.go:543-550, is inlined into it): every evaluator call on the way with the level and scale
 * of its ciphertext arguments and results and its scalar arguments. Nothing is planted, nothing is digested (the run's keys are random);
 * what is pinned is the STAGE STRUCTURE: which op at which level with which scale / constant. Hooks sit behind the stack check. */
typedef struct { const char *name; uint64_t fn; int in_ct[3]; int in_iface[2]; int f64[2]; int i64[2]; int out_arg; int out_res[2]; int out_slice; } flow_t;
#define NA (-1)
static flow_t g_flow[] = {
  /* name                       fn         in_ct (entry rsp +)   operands by iface(data) f64 args   int args   ctOut arg  results (entry rsp +)  []*ct result */
  {"BootstrappConv_CtoS",       0x506800, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, 0x20},          NA},
  {"Bootstrapp",                0x505ce0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, NA},            NA},
  {"modUp",                     0x507400, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, NA},            NA},
  {"CoeffsToSlots",             0x507dc0, {0x08, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},NA,        {0x38, 0x40},          NA},
  {"SlotsToCoeffs",             0x508140, {0x08, 0x10, NA},      {NA, NA},              {NA, NA},  {0x20, NA},NA,        {0x40, NA},            NA},
  {"evaluateSine",              0x508380, {0x10, 0x18, NA},      {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x20, 0x28},          NA},
  {"evaluateCheby",             0x508540, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, NA},            NA},
  {"EvaluateCheby",             0x52d7c0, {0x10, NA, NA},        {NA, NA},              {0x20, NA},{NA, NA},  NA,        {0x28, NA},            NA},
  {"EvaluatePoly",              0x52d3c0, {0x10, NA, NA},        {NA, NA},              {0x20, NA},{NA, NA},  NA,        {0x28, NA},            NA},
  {"LinearTransform",           0x5264c0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x20, NA},NA,        {NA, NA},              0x28},
  {"MultByConst",               0x51ee80, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, 0x20}, 0x28,   {NA, NA},              NA},
  {"AddConst",                  0x51d9e0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, 0x20}, 0x28,   {NA, NA},              NA},
  {"Rescale",                   0x522440, {0x10, NA, NA},        {NA, NA},              {0x18, NA},{NA, NA},  0x20,      {NA, NA},              NA},
  {"DropLevel",                 0x5223a0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},0x10,      {NA, NA},              NA},
  {"SetScale",                  0x521b80, {0x10, NA, NA},        {NA, NA},              {0x18, NA},{NA, NA},  0x10,      {NA, NA},              NA},
  {"mulRelin",                  0x522c60, {NA, NA, NA},          {0x18, 0x28},          {NA, NA},  {0x30, NA},0x38,      {NA, NA},              NA},
  {"Conjugate",                 0x5247e0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  0x18,      {NA, NA},              NA},
  {"ConjugateNew",              0x524680, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, NA},            NA},
  {"MultByi",                   0x520b00, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  0x18,      {NA, NA},              NA},
  {"MultByiNew",                0x5209a0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  NA,        {0x18, NA},            NA},
  {"DivByi",                    0x521280, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  0x18,      {NA, NA},              NA},
  {"Add",                       0x51aee0, {NA, NA, NA},          {0x18, 0x28},          {NA, NA},  {NA, NA},  0x30,      {NA, NA},              NA},
  {"AddNew",                    0x51b1a0, {NA, NA, NA},          {0x18, 0x28},          {NA, NA},  {NA, NA},  NA,        {0x30, NA},            NA},
  {"Sub",                       0x51b320, {NA, NA, NA},          {0x18, 0x28},          {NA, NA},  {NA, NA},  0x30,      {NA, NA},              NA},
  {"SubNew",                    0x51b9a0, {NA, NA, NA},          {0x18, 0x28},          {NA, NA},  {NA, NA},  NA,        {0x30, NA},            NA},
  {"Neg",                       0x51d260, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {NA, NA},  0x18,      {NA, NA},              NA},
  {"MulByPow2",                 0x521e80, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},0x20,      {NA, NA},              NA},
  {"MulByPow2New",              0x521d00, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},NA,        {0x20, NA},            NA},
  {"ScaleUp",                   0x521ac0, {0x10, NA, NA},        {NA, NA},              {0x18, NA},{NA, NA},  0x20,      {NA, NA},              NA},
  {"Rotate",                    0x524420, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},0x20,      {NA, NA},              NA},
  {"RotateNew",                 0x5242a0, {0x10, NA, NA},        {NA, NA},              {NA, NA},  {0x18, NA},NA,        {0x20, NA},            NA},
  {"main.evalReLU",             0x53a040, {NA, NA, NA},          {NA, NA},              {NA, NA},  {NA, NA},  NA,        {NA, NA},              NA},
  {"main.keep_ctxt",            0x5399c0, {NA, NA, NA},          {NA, NA},              {NA, NA},  {NA, NA},  NA,        {NA, NA},              NA},
  {"main.evalConv_BNRelu_new",  0x53d440, {NA, NA, NA},          {NA, NA},              {NA, NA},  {NA, NA},  NA,        {NA, NA},              NA},
};
static int g_flow_on = 0, g_flow_mode = 0, g_flow_depth = 0;
/* -chain: -flow plus planted data and digests, end to end: the input of BootstrappConv_CtoS is planted (SEED_OPIN(4000, 0, poly, limb)), every
 * switching key is planted by the KIND of key switch that reads it - SwitchKeysInPlace (relinearisation, conjugation): id CH_SWITCH_ID,
 * KeyswitchHoistedNoModDown (baby steps): LT_BABY_ID, SwitchKeysInPlaceNoModDown (giant steps): LT_GIANT_ID - and the results of
 * BootstrappConv_CtoS, CoeffsToSlots, evaluateSine, SlotsToCoeffs and the Rescale behind it are digested. */
#define CH_SWITCH_ID 42
static int g_chain = 0, g_flow_bl = 0;
static void lt_plant_key(uint64_t level, uint64_t evk, int id);
static void on_ch_switch(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_flow_on) return; lt_plant_key(rd64(r->rsp + 0x10), rd64(r->rsp + 0x20), CH_SWITCH_ID); }
static void on_ch_baby(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_flow_on) return; lt_plant_key(rd64(r->rsp + 0x10), rd64(r->rsp + 0x48), 40); }
static void on_ch_giant(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_flow_on) return; lt_plant_key(rd64(r->rsp + 0x10), rd64(r->rsp + 0x20), 41); }
static uint64_t post_check(uint64_t fn) {            /* first instruction behind Go's stack check (function start if it has none) */
    uint8_t b[40]; rd(fn, b, sizeof b);
    if (!(b[0] == 0x64 && b[1] == 0x48 && b[2] == 0x8b)) return fn;
    for (int i = 9; i < 30; i++) if (b[i] == 0x3b && (b[i+1] == 0x61 || b[i+1] == 0x41) && b[i+2] == 0x10) {
        if (b[i+3] == 0x0f && b[i+4] == 0x86) return fn + (uint64_t)i + 9;
        if (b[i+3] == 0x76) return fn + (uint64_t)i + 5;
    }
    fprintf(stderr, "no stack check found at %#lx\n", fn); exit(3);
}
static int plausible_ct(uint64_t p) {                  /* heap pointer whose first word points at a []*ring.Poly header of length 2 or 3 */
    if (p < 0xc000000000ull || p > 0xd000000000ull) return 0;
    uint64_t inner = rd64(p); if (inner < 0xc000000000ull || inner > 0xd000000000ull) return 0;
    uint64_t n = rd64(inner + 8); return n >= 1 && n <= 3 && rd64(inner) >= 0xc000000000ull;
}
static void flow_ct(const char *key, int idx, uint64_t ct) {
    if (!plausible_ct(ct)) { fprintf(g_out, ", \"%s%d\": null", key, idx); return; }
    fprintf(g_out, ", \"%s%d\": {\"level\": %d, \"scale\": %.17g, \"degree\": %d}", key, idx, poly_limbs(ct_poly(ct, 0)) - 1, ct_scale(ct), ct_degree1(ct) - 1);
}
typedef struct { flow_t *f; uint64_t out_arg; int depth; } flowret_t;
#define MAXFLOWRET 4096
static flowret_t g_flowret[MAXFLOWRET]; static int g_flowret_i;
static void flow_done(void) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); }
static void ret_flow(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; flowret_t *u = ud; flow_t *f = u->f;
    uint64_t E = r->rsp - 8;                           /* the entry stack pointer */
    emit_begin("ret"); fprintf(g_out, ", \"fn\": \"%s\", \"depth\": %d", f->name, u->depth);
    if (u->out_arg) flow_ct("out", 0, u->out_arg);
    for (int i = 0; i < 2; i++) if (f->out_res[i] != NA) flow_ct("res", i, rd64(E + (uint64_t)f->out_res[i]));
    if (f->out_slice != NA) { uint64_t p = rd64(E + (uint64_t)f->out_slice), n = rd64(E + (uint64_t)f->out_slice + 8); for (uint64_t i = 0; i < n && i < 2; i++) flow_ct("res", (int)i, rd64(p + 8 * i)); }
    if (g_chain) {
        const int want = !strcmp(f->name, "BootstrappConv_CtoS") || !strcmp(f->name, "Bootstrapp") || (!strcmp(f->name, "SetScale") && u->depth <= 1) || !strcmp(f->name, "CoeffsToSlots") || !strcmp(f->name, "evaluateSine") || !strcmp(f->name, "SlotsToCoeffs") ||
                         !strcmp(f->name, "LinearTransform") || !strcmp(f->name, "ConjugateNew") || !strcmp(f->name, "modUp") || !strcmp(f->name, "EvaluateCheby") || !strcmp(f->name, "EvaluatePoly") ||
                         (!strcmp(f->name, "Rescale") && u->depth <= 2) || (!strcmp(f->name, "MultByConst") && u->depth <= 1) || (!strcmp(f->name, "mulRelin") && u->depth <= 1);
        if (want) {
            if (u->out_arg && plausible_ct(u->out_arg)) emit_ct("digest_out", u->out_arg);
            for (int i = 0; i < 2; i++) if (f->out_res[i] != NA) { uint64_t c = rd64(E + (uint64_t)f->out_res[i]); char key[16]; snprintf(key, sizeof key, "digest_res%d", i); if (plausible_ct(c)) emit_ct(key, c); }
            if (f->out_slice != NA) { uint64_t p = rd64(E + (uint64_t)f->out_slice), n = rd64(E + (uint64_t)f->out_slice + 8); if (n && plausible_ct(rd64(p))) emit_ct("digest_res0", rd64(p)); }
        }
    }
    emit_end(); g_flow_depth = u->depth;
    if (!strcmp(f->name, "main.evalConv_BNRelu_new") || (g_flow_bl && !strcmp(f->name, "Bootstrapp"))) flow_done(); }
static void on_flow(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; flow_t *f = ud;
    if (!g_flow_mode) return;
    if (!g_flow_on) {                                   /* the enclosing layer function is hooked from its entry so that its return ends the run */
        if (!strcmp(f->name, "main.evalConv_BNRelu_new")) { flowret_t *u0 = &g_flowret[g_flowret_i++ % MAXFLOWRET]; u0->f = f; u0->depth = 0; u0->out_arg = 0; hook_return(r, ret_flow, u0); return; }
        if (g_flow_bl) { if (strcmp(f->name, "Bootstrapp")) return; g_flow_on = 1; g_flow_depth = 0;                    /* -flow-bl: the stock Bootstrapp of the baseline half, entry to return */
            if (g_chain) plant_ct(rd64(r->rsp + (uint64_t)f->in_ct[0]), 4001, 0);                                            /* with -chain: planted input (both limbs of the level-1 ciphertext), planted keys, digests */
            goto log_call; }
        if (strcmp(f->name, "BootstrappConv_CtoS")) return;
        g_flow_on = 1; g_flow_depth = 1;
        if (g_chain) { uint64_t ct = rd64(r->rsp + 0x10); if (poly_limbs(ct_poly(ct, 0)) != 1) { fprintf(stderr, "-chain expects a level-0 input\n"); exit(3); } plant_ct(ct, 4000, 0); } }
log_call:;
    uint64_t E = r->rsp;
    emit_begin("call"); fprintf(g_out, ", \"fn\": \"%s\", \"depth\": %d", f->name, g_flow_depth);
    for (int i = 0; i < 3; i++) if (f->in_ct[i] != NA) flow_ct("in", i, rd64(E + (uint64_t)f->in_ct[i]));
    for (int i = 0; i < 2; i++) if (f->in_iface[i] != NA) flow_ct("op", i, rd64(E + (uint64_t)f->in_iface[i]));
    for (int i = 0; i < 2; i++) if (f->f64[i] != NA) fprintf(g_out, ", \"f%d\": %.17g", i, rdf64(E + (uint64_t)f->f64[i]));
    for (int i = 0; i < 2; i++) if (f->i64[i] != NA) fprintf(g_out, ", \"i%d\": %ld", i, (long)rd64(E + (uint64_t)f->i64[i]));
    if (!strcmp(f->name, "MultByConst") || !strcmp(f->name, "AddConst")) {      /* constant interface{}: type word + data word */
        uint64_t ty = rd64(E + 0x18), data = rd64(E + 0x20); uint64_t w0 = 0, w1 = 0;
        if (data >= 0x400000) { w0 = rd64(data); w1 = rd64(data + 8); }
        double d0, d1; memcpy(&d0, &w0, 8); memcpy(&d1, &w1, 8);
        fprintf(g_out, ", \"const_type\": %lu, \"const_words\": [%lu, %lu], \"const_f64\": [%.17g, %.17g]", ty, w0, w1, d0, d1);
    }
    if (!strcmp(f->name, "LinearTransform")) {           /* *PtDiagMatrix: LogSlots, N1, Level, Scale */
        uint64_t m = rd64(E + 0x20);
        if (m >= 0xc000000000ull) fprintf(g_out, ", \"matrix\": {\"ptr\": %lu, \"LogSlots\": %lu, \"N1\": %lu, \"Level\": %lu, \"Scale\": %.17g}", m, rd64(m), rd64(m + 8), rd64(m + 16), rdf64(m + 24));
    }
    if (!strcmp(f->name, "EvaluateCheby")) {             /* *ChebyshevInterpolation{Poly{maxDeg, coeffs, lead}, a, b} */
        uint64_t c = rd64(E + 0x18);
        fprintf(g_out, ", \"cheby\": {\"maxDeg\": %lu, \"ncoeffs\": %lu, \"lead\": %lu}", rd64(c), rd64(c + 16), rd64(c + 32) & 0xff);
    }
    emit_end();
    flowret_t *u = &g_flowret[g_flowret_i++ % MAXFLOWRET]; u->f = f; u->depth = g_flow_depth++;
    u->out_arg = f->out_arg != NA ? rd64(E + (uint64_t)f->out_arg) : 0;
    hook_return(r, ret_flow, u); }

/* -lt N: ckks.(*evaluator).LinearTransform (-> MultiplyByDiagMatrixBSGS) on planted data: at the entry of the first N calls inside the
 * convReLU chain the input ciphertext is planted (SEED_OPIN(3000 + call, 0, poly, limb)); every rotation key a nested
 * KeyswitchHoistedNoModDown (baby steps) reads is planted with key identity LT_BABY_ID, every key a nested SwitchKeysInPlaceNoModDown
 * (giant steps) reads with LT_GIANT_ID - the SAME rows for every baby key and for every giant key, because Go walks the giant steps in
 * map order and the tracer cannot tell which rotation a key belongs to; the algorithm does not care what the key rows are. The
 * plaintext diagonals are the run's own (pinned by -diag). Recorded: the matrix header, per nested ModDownSplitNTTPQ the digests of its
 * inputs (mod Q, mod P) and output, per SwitchKeysInPlaceNoModDown the digest of its input, and the returned ciphertext. */
#define LT_BABY_ID 40
#define LT_GIANT_ID 41
static int g_lt_max = 0, g_lt_calls = 0, g_in_lt = 0;
static void lt_plant_key(uint64_t level, uint64_t evk, int id) {
    int alpha = g_nP; { uint64_t v0 = rd64(evk); int limbs0 = poly_limbs(rd64(v0)); if (g_nQ_full) alpha = limbs0 - g_nQ_full; }
    const int beta = ((int)level + 1 + alpha - 1) / alpha; uint64_t v = rd64(evk);
    for (int d = 0; d < beta; d++) for (int k = 0; k < 2; k++) {
        uint64_t poly = rd64(v + 16ull * (uint64_t)d + 8ull * (uint64_t)k); int limbs = poly_limbs(poly);
        for (int l = 0; l <= (int)level; l++) plant_row(poly_row(poly, l, NULL), SEED_KSEVK(id, d, k, l), g_Q[l]);
        for (int j = 0; j < alpha; j++) plant_row(poly_row(poly, limbs - alpha + j, NULL), SEED_KSEVK(id, d, k, 32 + j), g_Pm[j]);
    }
}
/* -blop N (with -keep-bl -noplant, `conv k i 1`): the BASELINE operator main.evalConv_BN_BL_test (eval.go:78-134: preConv_BL's hoisted input rotations, the
 * k^2 plaintext products per output rotation, the output rotations, the BN bias) as a whole on planted data: at the entry of the first N calls the input
 * ciphertext is planted (SEED_OPIN(5000 + call, 0, poly, limb)), every rotation key its KeyswitchHoisted (input rotations) reads is planted with identity
 * LT_BABY_ID and every key a RotateNew's SwitchKeysInPlaceNoModDown reads with LT_GIANT_ID (the same rows for every key of a kind), and the returned
 * ciphertext is digested. The plaintexts are the run's own (kernel CSVs through the pinned slot encoder). Args (Go 1.16 stack ABI): cont, ct_input, ker_in,
 * bn_a, bn_b (3 slices), seven ints, two bools: 0x98 bytes from [rsp+8]; the result pointer follows at [rsp+0xa0]. */
static int g_blop_max = 0, g_blop_calls = 0, g_in_blop = 0;
static void ret_blop(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t E = r->rsp - 8, ct = rd64(E + 0xa0);
    g_in_blop = 0;
    emit_begin("evalConv_BN_BL_test.ret"); fprintf(g_out, ", \"call\": %d", g_blop_calls - 1); emit_ct("out", ct); emit_end();
    if (g_blop_calls >= g_blop_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); } }
static void on_blop(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_blop_max || g_blop_calls >= g_blop_max) return;
    uint64_t ct = rd64(r->rsp + 0x10); const int call = g_blop_calls++;
    plant_ct(ct, 5000 + call, 0);
    g_in_blop = 1;
    emit_begin("evalConv_BN_BL_test.call"); fprintf(g_out, ", \"call\": %d, \"in_wid\": %lu, \"ker_wid\": %lu, \"real_ib\": %lu, \"real_ob\": %lu, \"pos\": %lu, \"norm\": %lu, \"pad\": %lu",
        call, rd64(r->rsp + 0x60), rd64(r->rsp + 0x68), rd64(r->rsp + 0x70), rd64(r->rsp + 0x78), rd64(r->rsp + 0x80), rd64(r->rsp + 0x88), rd64(r->rsp + 0x90));
    emit_ct("in", ct); emit_end();
    hook_return(r, ret_blop, ud_new(0, 0, 0)); }
static void blop_plant(uint64_t level, uint64_t evk, int id, const char *what) {
    const uint64_t v0 = rd64(evk); const int limbs0 = poly_limbs(rd64(v0));
    if ((int)level >= g_nQ || limbs0 - g_nQ_full < 1 || limbs0 - g_nQ_full > g_nP) { fprintf(stderr, "-blop: %s at level %lu with %d-limb key rows does not fit -Q / -P / -nq-full\n", what, level, limbs0); exit(3); }
    lt_plant_key(level, evk, id); }
static void on_blop_hoisted(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_blop) return; blop_plant(rd64(r->rsp + 0x10), rd64(r->rsp + 0x48), LT_BABY_ID, "KeyswitchHoistedNoModDown"); }
static void on_blop_switch(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_blop) return; blop_plant(rd64(r->rsp + 0x10), rd64(r->rsp + 0x20), LT_GIANT_ID, "SwitchKeysInPlaceNoModDown"); }
static void on_lt_ks_hoisted(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_lt) return;
    uint64_t level = rd64(r->rsp + 0x10), evk = rd64(r->rsp + 0x48);
    lt_plant_key(level, evk, LT_BABY_ID);
    emit_begin("lt.KeyswitchHoistedNoModDown"); fprintf(g_out, ", \"level\": %lu", level); emit_end(); }
static void on_lt_ks_nomoddown(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_lt) return;
    uint64_t level = rd64(r->rsp + 0x10), cx = rd64(r->rsp + 0x18), evk = rd64(r->rsp + 0x20);
    lt_plant_key(level, evk, LT_GIANT_ID);
    emit_begin("lt.SwitchKeysInPlaceNoModDown"); fprintf(g_out, ", \"level\": %lu", level); emit_poly("cx", cx, (int)level + 1); emit_end(); }
static void ret_lt_moddown(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r; ud3_t *u = ud; emit_poly("out", u->a, (int)u->b + 1); emit_end(); }
static void on_lt_moddown(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_lt) return;
    uint64_t level = rd64(r->rsp + 0x10), pq = rd64(r->rsp + 0x18), pp = rd64(r->rsp + 0x20), out = rd64(r->rsp + 0x28);
    emit_begin("lt.ModDownSplitNTTPQ"); fprintf(g_out, ", \"level\": %lu", level); emit_poly("inQ", pq, (int)level + 1); emit_poly("inP", pp, poly_limbs(pp));
    hook_return(r, ret_lt_moddown, ud_new(out, level, 0)); }
static void ret_lt(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t E = r->rsp - 8, p = rd64(E + 0x28), n = rd64(E + 0x30);
    g_in_lt = 0;
    emit_begin("LinearTransform.end"); fprintf(g_out, ", \"call\": %d, \"nres\": %lu", g_lt_calls - 1, n); if (n) emit_ct("out", rd64(p)); emit_end();
    if (g_lt_calls >= g_lt_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); } }
static void on_lt(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_lt_max || g_in_lt || g_lt_calls >= g_lt_max) return;
    uint64_t ct0 = rd64(r->rsp + 0x10), m = rd64(r->rsp + 0x20);
    int level = poly_limbs(ct_poly(ct0, 0)) - 1, call = g_lt_calls++;
    if (level >= g_nQ) { fprintf(stderr, "LinearTransform at level %d: pass the modulus chain with -Q\n", level); exit(3); }
    plant_ct(ct0, 3000 + call, 0);
    emit_begin("LinearTransform.begin"); fprintf(g_out, ", \"call\": %d, \"level\": %d, \"scale_in\": %.17g, \"matrix\": {\"LogSlots\": %lu, \"N1\": %lu, \"Level\": %lu, \"Scale\": %.17g}",
        call, level, ct_scale(ct0), rd64(m), rd64(m + 8), rd64(m + 16), rdf64(m + 24));
    emit_ct("in", ct0); emit_end();
    g_in_lt = 1;
    fprintf(stderr, "LinearTransform call %d level %d\n", call, level);
    hook_return(r, ret_lt, NULL); }

/* -poly N: ckks.(*evaluator).EvaluatePoly one level up (conv.go:460-477: the three sign polynomials of evalReLU). At the entry of the
 * first N calls the input ciphertext is planted (SEED_OPIN(call, 0, poly, limb) mod q_limb); every relinearisation key a nested
 * SwitchKeysInPlace reads is planted as in -ks / -ops (SEED_KSEVK by key identity). Recorded: the polynomial (maxDeg, lead, the real
 * parts of its coefficients), targetScale, and -- log only -- every nested computePowerBasis / recurse / evaluatePolyFromPowerBasis /
 * mulRelin / Rescale / MultByGaussianIntegerAndAdd / AddConst / DropLevel with its arguments, levels, scales and the SHA-256 of each
 * ciphertext result, then the returned ciphertext. The oracle replays the polynomial on the same planted data and must reproduce
 * every digest. */
static int g_poly_max = 0, g_poly_calls = 0, g_in_poly = 0, g_cheby = 0;   /* -cheby N: the same trace over EvaluateCheby (the sine of evaluateSine) */
static int g_in_poly_fwd(void) { return g_in_poly; }
static void emit_polyarg(const char *key, uint64_t pol) {
    uint64_t maxdeg = rd64(pol), cp = rd64(pol + 8), cl = rd64(pol + 16); uint8_t lead; rd(pol + 32, &lead, 1);
    fprintf(g_out, ", \"%s\": {\"maxDeg\": %lu, \"lead\": %d, \"coeffs\": [", key, maxdeg, (int)lead);
    for (uint64_t i = 0; i < cl; i++) { double c[2]; rd(cp + 16 * i, c, 16); fprintf(g_out, "%s[%.17g, %.17g]", i ? ", " : "", c[0], c[1]); }
    fprintf(g_out, "]}");
}
typedef struct { uint64_t out; int id; } prec_t;
static prec_t g_prec[64]; static int g_prec_i;
static prec_t *prec_new(uint64_t out) { prec_t *u = &g_prec[g_prec_i++ % 64]; u->out = out; u->id = 0; return u; }
static void ret_p_ct(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)r; prec_t *u = ud; emit_ct("out", u->out); emit_end(); }
static void on_p_mulrelin(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t op0 = rd64(r->rsp + 0x18), op1 = rd64(r->rsp + 0x28), out = rd64(r->rsp + 0x38); uint8_t relin; rd(r->rsp + 0x30, &relin, 1);
    emit_begin("p.mulRelin"); fprintf(g_out, ", \"relin\": %d, \"level0\": %d, \"level1\": %d, \"scale0\": %.17g, \"scale1\": %.17g, \"square\": %d", (int)relin,
        poly_limbs(ct_poly(op0, 0)) - 1, poly_limbs(ct_poly(op1, 0)) - 1, ct_scale(op0), ct_scale(op1), op0 == op1);
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_rescale(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t in = rd64(r->rsp + 0x10), out = rd64(r->rsp + 0x20); double ms = rdf64(r->rsp + 0x18);
    emit_begin("p.Rescale"); fprintf(g_out, ", \"level_in\": %d, \"scale_in\": %.17g, \"min_scale\": %.17g", poly_limbs(ct_poly(in, 0)) - 1, ct_scale(in), ms);
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_mgiaa(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t in = rd64(r->rsp + 0x10), out = rd64(r->rsp + 0x28); int64_t cr = (int64_t)rd64(r->rsp + 0x18), ci = (int64_t)rd64(r->rsp + 0x20);
    emit_begin("p.MultByGaussianIntegerAndAdd"); fprintf(g_out, ", \"cReal\": %ld, \"cImag\": %ld, \"level_in\": %d, \"scale_in\": %.17g, \"level_out\": %d, \"scale_out\": %.17g",
        (long)cr, (long)ci, poly_limbs(ct_poly(in, 0)) - 1, ct_scale(in), poly_limbs(ct_poly(out, 0)) - 1, ct_scale(out));
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_addconst(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t in = rd64(r->rsp + 0x10), ty = rd64(r->rsp + 0x18), data = rd64(r->rsp + 0x20), out = rd64(r->rsp + 0x28); double c[2] = {0, 0};
    if (ty == A_TYPE_FLOAT64) c[0] = rdf64(data); else rd(data, c, 16);
    emit_begin("p.AddConst"); fprintf(g_out, ", \"type\": %lu, \"re\": %.17g, \"im\": %.17g, \"level\": %d, \"scale\": %.17g", ty, c[0], c[1], poly_limbs(ct_poly(in, 0)) - 1, ct_scale(in));
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_droplevel(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t in = rd64(r->rsp + 0x10);
    emit_begin("p.DropLevel"); fprintf(g_out, ", \"levels\": %lu, \"level_in\": %d", rd64(r->rsp + 0x18), poly_limbs(ct_poly(in, 0)) - 1); emit_end(); }
static void on_p_recurse(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    emit_begin("p.recurse"); fprintf(g_out, ", \"targetScale\": %.17g, \"logSplit\": %lu, \"logDegree\": %lu", rdf64(r->rsp + 8), rd64(r->rsp + 0x10), rd64(r->rsp + 0x18));
    emit_polyarg("coeffs", rd64(r->rsp + 0x20)); emit_end(); }
static void on_p_leaf(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    emit_begin("p.evaluatePolyFromPowerBasis"); fprintf(g_out, ", \"targetScale\": %.17g", rdf64(r->rsp + 8)); emit_polyarg("coeffs", rd64(r->rsp + 0x10)); emit_end(); }
static void on_p_powerbasis(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    emit_begin("p.computePowerBasis"); fprintf(g_out, ", \"n\": %lu, \"scale\": %.17g", rd64(r->rsp + 8), rdf64(r->rsp + 0x18)); emit_end(); }
static void on_p_add(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t op0 = rd64(r->rsp + 0x18), op1 = rd64(r->rsp + 0x28), out = rd64(r->rsp + 0x30);
    emit_begin("p.Add"); fprintf(g_out, ", \"level0\": %d, \"level1\": %d, \"scale0\": %.17g, \"scale1\": %.17g, \"out_is_op0\": %d", poly_limbs(ct_poly(op0, 0)) - 1, poly_limbs(ct_poly(op1, 0)) - 1,
        ct_scale(op0), ct_scale(op1), out == op0);
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_sub(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t op0 = rd64(r->rsp + 0x18), op1 = rd64(r->rsp + 0x28), out = rd64(r->rsp + 0x30);
    emit_begin("p.Sub"); fprintf(g_out, ", \"level0\": %d, \"level1\": %d, \"scale0\": %.17g, \"scale1\": %.17g, \"out_is_op0\": %d", poly_limbs(ct_poly(op0, 0)) - 1, poly_limbs(ct_poly(op1, 0)) - 1,
        ct_scale(op0), ct_scale(op1), out == op0);
    hook_return(r, ret_p_ct, prec_new(out)); }
static void on_p_multbyconst(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud; if (!g_in_poly) return;
    uint64_t in = rd64(r->rsp + 0x10), ty = rd64(r->rsp + 0x18), data = rd64(r->rsp + 0x20), out = rd64(r->rsp + 0x28); uint64_t raw = rd64(data);
    emit_begin("p.MultByConst"); fprintf(g_out, ", \"type\": %lu, \"is_f64\": %d, \"raw_u64\": %lu, \"as_f64\": %.17g, \"level\": %d, \"scale\": %.17g", ty, ty == A_TYPE_FLOAT64, raw, rdf64(data),
        poly_limbs(ct_poly(in, 0)) - 1, ct_scale(in));
    hook_return(r, ret_p_ct, prec_new(out)); }
static void ret_evalpoly(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t out = rd64(r->rsp - 8 + 0x28);
    g_in_poly = 0; g_nested_ks = 0;
    emit_begin("EvaluatePoly.end"); fprintf(g_out, ", \"call\": %d, \"err\": %lu", g_poly_calls - 1, rd64(r->rsp - 8 + 0x30)); if (out) emit_ct("out", out); emit_end();
    if (g_poly_calls >= g_poly_max) { fprintf(g_out, "\n ],\n \"exit_code\": 0}\n"); fflush(g_out); kill(g_pid, SIGKILL); exit(0); } }
static void on_evalpoly(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (!g_poly_max || g_in_poly || g_poly_calls >= g_poly_max) return;
    uint64_t ct0 = rd64(r->rsp + 0x10), pol = rd64(r->rsp + 0x18); double ts = rdf64(r->rsp + 0x20);
    int level = poly_limbs(ct_poly(ct0, 0)) - 1, call = g_poly_calls++;
    if (level >= g_nQ) { fprintf(stderr, "EvaluatePoly at level %d: pass the modulus chain with -Q\n", level); exit(3); }
    plant_ct(ct0, 1000 + call, 0);
    emit_begin("EvaluatePoly.begin"); fprintf(g_out, ", \"call\": %d, \"level\": %d, \"scale_in\": %.17g, \"targetScale\": %.17g", call, level, ct_scale(ct0), ts);
    emit_polyarg("pol", pol);
    if (g_cheby) { double ab[4]; rd(pol + 40, ab, 32); fprintf(g_out, ", \"a\": %.17g, \"b\": %.17g", ab[0], ab[2]); }
    emit_ct("in", ct0); emit_end();
    g_in_poly = 1; g_nested_ks = 1; g_nested_evk = -1;
    fprintf(stderr, "EvaluatePoly call %d level %d scale %g target %g\n", call, level, ct_scale(ct0), ts);
    hook_return(r, ret_evalpoly, NULL); }

/* conv_then_pack(params (0x68 bytes by value), pack_evaluator (itab,ptr), ctxt_in, pl_ker (ptr,len,cap),
 *                plain_idx (ptr,len,cap), max_ob, norm, ECD_LV int, out_scale float64) *Ciphertext
 * entry-rsp offsets (from the frame layout of test_run:main.conv_then_pack, sub $0x150 / args at 0x158):
 *   ctxt_in +0x80, pl_ker +0x88/+0x90, plain_idx +0xa0/+0xa8, max_ob +0xb8, norm +0xc0, ECD_LV +0xc8,
 *   out_scale +0xd0, result +0xd8 */
static void ret_ctp(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    uint64_t ct = rd64(r->rsp - 8 + 0xd8);
    emit_begin("conv_then_pack.return"); emit_ct("out", ct); emit_end();
    g_in_ctp = 0; g_after_ctp = 1;
}
static int g_ctp_calls;
static void on_ctp(pid_t t, struct user_regs_struct *r, void *ud) { (void)t; (void)ud;
    if (g_diag_max) { on_ctp_diag(t, r, ud); return; }
    if (g_ks_max) return;
    uint64_t E = r->rsp;
    uint64_t ct_in = rd64(E + 0x80), ker = rd64(E + 0x88), nker = rd64(E + 0x90), idx = rd64(E + 0xa0), nidx = rd64(E + 0xa8);
    uint64_t max_ob = rd64(E + 0xb8), norm = rd64(E + 0xc0), ecd = rd64(E + 0xc8); double out_scale = rdf64(E + 0xd0);
    fprintf(stderr, "conv_then_pack: ct_in=%#lx pl_ker=%#lx(%lu) idx=%#lx(%lu) max_ob=%lu norm=%lu ECD_LV=%lu out_scale=%g\n",
            ct_in, ker, nker, idx, nidx, max_ob, norm, ecd, out_scale);
    if (g_mode_probe) {
        dump_words("ct_in", ct_in, 4); dump_words(" *ct_in[0] (rlwe.Ciphertext)", rd64(ct_in), 4);
        uint64_t p0 = ct_poly(ct_in, 0); dump_words("  poly0", p0, 6); dump_words("   hdrs", rd64(p0), 6);
        uint64_t pt0 = rd64(ker); dump_words("pl_ker[0]", pt0, 4); dump_words(" *pl_ker[0][0] (rlwe.Plaintext)", rd64(pt0), 4);
        dump_words("  poly", pt_poly(pt0), 6);
        fprintf(stderr, "ct_in scale=%g deg+1=%d limbs=%d; pt scale=%g limbs=%d\n", ct_scale(ct_in), ct_degree1(ct_in),
                poly_limbs(ct_poly(ct_in, 0)), pt_scale(pt0), poly_limbs(pt_poly(pt0)));
        uint64_t i0 = rd64(idx); fprintf(stderr, "idx[0] scale=%g limbs=%d\n", pt_scale(i0), poly_limbs(pt_poly(i0)));
    }
    if (g_ctp_calls++ > 0) return;          /* first call only */
    g_in_ctp = 1;
    emit_begin("conv_then_pack.entry");
    fprintf(g_out, ", \"max_ob\": %lu, \"norm\": %lu, \"ECD_LV\": %lu, \"out_scale\": %.17g, \"ct_in_scale\": %.17g, \"pl_ker_scale\": %.17g",
            max_ob, norm, ecd, out_scale, ct_scale(ct_in), pt_scale(rd64(ker)));
    emit_end();
    /* digests of the reference's own pl_ker[i] (prep_Ker output, conv.go:510-515) before they are overwritten */
    for (uint64_t i = 0; i < nker && !(g_lean && i >= 4); i++) {
        uint64_t pt = rd64(ker + 8*i);
        emit_begin("pl_ker_orig"); fprintf(g_out, ", \"i\": %lu, \"scale\": %.17g", i, pt_scale(pt));
        emit_poly("pt", pt_poly(pt), poly_limbs(pt_poly(pt))); emit_end();
    }
    /* plant ct_in and pl_ker */
    for (int p = 0; p < 2; p++) {
        uint64_t poly = ct_poly(ct_in, p);
        plant_row(poly_row(poly, 0, NULL), SEED_CT(p, 0), Q0);
        plant_row(poly_row(poly, 1, NULL), SEED_CT(p, 1), Q1);
    }
    for (uint64_t i = 0; i < nker; i++) {
        uint64_t poly = pt_poly(rd64(ker + 8*i));
        plant_row(poly_row(poly, 0, NULL), SEED_KER(i, 0), Q0);
        plant_row(poly_row(poly, 1, NULL), SEED_KER(i, 1), Q1);
    }
    for (uint64_t s = 0; s < nidx; s++) {
        uint64_t pt = rd64(idx + 8*s);
        emit_begin("plain_idx"); fprintf(g_out, ", \"s\": %lu, \"scale\": %.17g", s, pt_scale(pt));
        emit_poly("pt", pt_poly(pt), poly_limbs(pt_poly(pt))); emit_end();
    }
    hook_return(r, ret_ctp, NULL);
}

static void usage(void) {
    fprintf(stderr, "usage: gotrace [-probe] [-seed S] [-o out.json] [-keep-bl] -- /root/reference/test_run conv K I N\n"); exit(2);
}

int main(int argc, char **argv) {
    const char *outpath = NULL; int ai = 1;
    for (; ai < argc; ai++) {
        if (!strcmp(argv[ai], "--")) { ai++; break; }
        else if (!strcmp(argv[ai], "-probe")) g_mode_probe = 1;
        else if (!strcmp(argv[ai], "-v")) g_verbose = 1;
        else if (!strcmp(argv[ai], "-lean")) g_lean = 1;
        else if (!strcmp(argv[ai], "-ks") && ai + 1 < argc) g_ks_max = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "-ops") && ai + 1 < argc) g_ops_max = atoi(argv[++ai]);            /* trace this many Rescale / mulRelin calls */
        else if (!strcmp(argv[ai], "-ops-unique") && ai + 1 < argc) g_ops_unique = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "-ks-unique") && ai + 1 < argc) g_ks_unique = atoi(argv[++ai]);   /* at most this many calls per (level, alpha) */
        else if (!strcmp(argv[ai], "-nq-full") && ai + 1 < argc) g_nQ_full = atoi(argv[++ai]);       /* Q limbs of a full key row: alpha = limbs - this */
        else if ((!strcmp(argv[ai], "-Q") || !strcmp(argv[ai], "-P")) && ai + 1 < argc) {
            int isq = argv[ai][1] == 'Q'; char *tok = strtok(argv[++ai], ",");
            while (tok) { if (isq) g_Q[g_nQ++] = strtoull(tok, NULL, 0); else g_Pm[g_nP++] = strtoull(tok, NULL, 0); tok = strtok(NULL, ","); }
        }
        else if (!strcmp(argv[ai], "-cheby") && ai + 1 < argc) { g_poly_max = atoi(argv[++ai]); g_cheby = 1; }
        else if (!strcmp(argv[ai], "-poly") && ai + 1 < argc) g_poly_max = atoi(argv[++ai]);          /* trace this many EvaluatePoly calls (planted input and keys) */
        else if (!strcmp(argv[ai], "-enc") && ai + 1 < argc) g_enc_max = atoi(argv[++ai]);            /* trace the slot encoder: this many invfft / Encode calls */
        else if (!strcmp(argv[ai], "-diag") && ai + 1 < argc) g_diag_max = atoi(argv[++ai]);          /* digest this many encoded DFT diagonals */
        else if (!strcmp(argv[ai], "-blop") && ai + 1 < argc) g_blop_max = atoi(argv[++ai]);          /* the baseline operator evalConv_BN_BL_test on planted input and keys */
        else if (!strcmp(argv[ai], "-logslots") && ai + 1 < argc) g_logslots = atoi(argv[++ai]);      /* with -diag: sparse-slot bootstrapper (see on_newbtp_mod) */
        else if (!strcmp(argv[ai], "-dump") && ai + 1 < argc) { g_dump = fopen(argv[++ai], "wb"); if (!g_dump) { perror("dump"); return 2; } }
        else if (!strcmp(argv[ai], "-flow")) g_flow_mode = 1;
        else if (!strcmp(argv[ai], "-chain")) { g_flow_mode = 1; g_chain = 1; }
        else if (!strcmp(argv[ai], "-flow-bl")) { g_flow_mode = 1; g_flow_bl = 1; g_skip_bl = 0; }
        else if (!strcmp(argv[ai], "-lt") && ai + 1 < argc) g_lt_max = atoi(argv[++ai]);               /* trace this many LinearTransform calls (planted input and rotation keys) */
        else if (!strcmp(argv[ai], "-keep-bl")) g_skip_bl = 0;
        else if (!strcmp(argv[ai], "-noplant")) g_noplant = 1;
        else if (!strcmp(argv[ai], "-seed") && ai + 1 < argc) g_seed = strtoull(argv[++ai], NULL, 0);
        else if (!strcmp(argv[ai], "-o") && ai + 1 < argc) outpath = argv[++ai];
        else usage();
    }
    if (ai >= argc) usage();
    g_out = outpath ? fopen(outpath, "w") : stdout;
    if (!g_out) { perror("fopen"); return 2; }
    g_tmp = malloc(g_N * 8);

    pid_t pid = fork();
    if (pid == 0) {
        ptrace(PTRACE_TRACEME, 0, 0, 0);
        execv(argv[ai], argv + ai);
        perror("execv"); _exit(127);
    }
    g_pid = pid; int st;
    waitpid(pid, &st, 0);
    if (!WIFSTOPPED(st)) { fprintf(stderr, "tracee did not stop\n"); return 3; }
    ptrace(PTRACE_SETOPTIONS, pid, 0, PTRACE_O_TRACECLONE | PTRACE_O_EXITKILL);
    char path[64]; snprintf(path, sizeof path, "/proc/%d/mem", pid);
    g_mem = open(path, O_RDWR);
    if (g_mem < 0) { perror("open mem"); return 3; }

    if (g_skip_bl) {   /* skip the BL baseline run (main.go:640); in tracee memory only */
        uint8_t cur[5], nop5[5] = {0x0f, 0x1f, 0x44, 0x00, 0x00};
        rd(A_CALL_BL, cur, 5);
        if (cur[0] != 0xe8) { fprintf(stderr, "unexpected byte at call site: %#x\n", cur[0]); return 3; }
        wr(A_CALL_BL, nop5, 5);
    }
    fprintf(g_out, "{\"seed\": %lu, \"N\": %lu, \"lean\": %d, \"moduli\": {\"Q0\": %lu, \"Q1\": %lu, \"P\": %lu},\n \"argv\": [", g_seed, g_N, g_lean, Q0, Q1, P0);
    for (int i = ai + 1; i < argc; i++) fprintf(g_out, "%s\"%s\"", i > ai + 1 ? ", " : "", argv[i]);
    fprintf(g_out, "],\n \"ks_Q\": ["); for (int i = 0; i < g_nQ; i++) fprintf(g_out, "%s%lu", i ? ", " : "", g_Q[i]);
    fprintf(g_out, "], \"ks_P\": ["); for (int i = 0; i < g_nP; i++) fprintf(g_out, "%s%lu", i ? ", " : "", g_Pm[i]);
    fprintf(g_out, "],\n \"events\": [");
    if (g_flow_mode) for (size_t i = 0; i < sizeof g_flow / sizeof g_flow[0]; i++) bp_add(post_check(g_flow[i].fn), on_flow, &g_flow[i]);
    if (g_chain) { bp_add(post_check(0x4fdd40), on_ch_switch, NULL); bp_add(post_check(0x4ff060), on_ch_baby, NULL); bp_add(post_check(0x4fe660), on_ch_giant, NULL); }
    if (g_logslots) { bp_add(post_check(0x510240), on_newbtp_mod, NULL); bp_add(post_check(0x50aa20), on_rots_for_btp, NULL); }
    if (g_flow_mode) goto hooks_done;          /* the flow hooks share addresses with the ones below (the first handler of an address wins) */
    bp_add(A_CONV_THEN_PACK, on_ctp, NULL);
    bp_add(A_ENCODECOEFFS, on_encode, NULL);
    bp_add(A_MULNEW, on_mulnew, NULL);
    bp_add(A_SETSCALE, on_setscale, NULL);
    bp_add(A_SUBNEW, on_subnew, NULL);
    bp_add(A_ADD, on_add, NULL);
    bp_add(A_ROTATEGAL, on_rotgal, NULL);
    bp_add(A_SWITCHKEYS, on_switchkeys, NULL);
    bp_add(A_MULTBYCONST, on_multbyconst, NULL);
    if (g_poly_max) { bp_add(g_cheby ? post_check(0x52d7c0) : A_EVALPOLY, on_evalpoly, NULL);
                      if (g_cheby) { bp_add(post_check(0x52f400), on_p_recurse, NULL); bp_add(post_check(0x52dee0), on_p_powerbasis, NULL); bp_add(post_check(0x51b320), on_p_sub, NULL); } bp_add(A_MULRELIN, on_p_mulrelin, NULL); bp_add(A_RESCALE, on_p_rescale, NULL); bp_add(A_MGIAA, on_p_mgiaa, NULL);
                      bp_add(A_ADDCONST, on_p_addconst, NULL); bp_add(A_DROPLEVEL, on_p_droplevel, NULL); bp_add(A_RECURSE, on_p_recurse, NULL); bp_add(A_POLYLEAF, on_p_leaf, NULL);
                      bp_add(A_POWERBASIS, on_p_powerbasis, NULL); bp_add(A_ADD, on_p_add, NULL); bp_add(A_MULTBYCONST, on_p_multbyconst, NULL); }
    if (g_lt_max) { bp_add(post_check(0x5264c0), on_lt, NULL); bp_add(post_check(0x4ff060), on_lt_ks_hoisted, NULL); bp_add(post_check(0x4fe660), on_lt_ks_nomoddown, NULL);
                    bp_add(post_check(0x4e4c40), on_lt_moddown, NULL); }
    if (g_blop_max) { bp_add(post_check(0x53bb80), on_blop, NULL); bp_add(post_check(0x4ff060), on_blop_hoisted, NULL); bp_add(post_check(0x4fe660), on_blop_switch, NULL); }
    if (g_diag_max) { bp_add(A_ENCDIAG, on_encdiag, NULL); bp_add(A_ENCMAT, on_encmat, NULL); }
    if (g_enc_max) { bp_add(A_INVFFT, on_invfft, NULL); bp_add(A_ENCODE, on_encode_slots, NULL); }
    if (g_ops_max) { bp_add(A_RESCALE, on_rescale, NULL); bp_add(A_MULRELIN, on_mulrelin, NULL); bp_add(A_ROTATE, on_rotate, NULL); bp_add(A_MODUP, on_modup, NULL); }
hooks_done:

    ptrace(PTRACE_CONT, pid, 0, 0);
    int exit_code = -1; uint64_t refire_addr = 0, refire_rsp = 0;
    for (;;) {
        pid_t t = waitpid(-1, &st, __WALL);
        if (t < 0) { if (errno == ECHILD) break; if (errno == EINTR) continue; perror("waitpid"); break; }
        if (WIFEXITED(st) || WIFSIGNALED(st)) {
            if (t == pid) { exit_code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st); }
            continue;
        }
        if (!WIFSTOPPED(st)) continue;
        int sig = WSTOPSIG(st);
        if (sig == SIGTRAP) {
            int ev = st >> 16;
            if (ev == PTRACE_EVENT_CLONE) { ptrace(PTRACE_CONT, t, 0, 0); continue; }
            struct user_regs_struct r; ptrace(PTRACE_GETREGS, t, 0, &r);
            bp_t *b = bp_find(r.rip - 1);
            if (b && b->armed) {
                r.rip -= 1;
                uint64_t addr = b->addr;
                handler_t fn = b->fn; void *ud = b->ud;
                if (refire_addr == addr && refire_rsp == r.rsp) refire_addr = 0;   /* same hit, handler already ran */
                else fn(t, &r, ud);                  /* may add/release breakpoints (including this one) */
                b = bp_find(addr);
                ptrace(PTRACE_SETREGS, t, 0, &r);
                if (b && b->armed) {                 /* step over, then re-arm */
                    bp_disarm(b);
                    ptrace(PTRACE_SINGLESTEP, t, 0, 0);
                    int st2; waitpid(t, &st2, __WALL);
                    if (WIFSTOPPED(st2) && WSTOPSIG(st2) != SIGTRAP) {
                        /* a signal (Go's SIGURG preemption) raced the step: re-arm and deliver it; if the
                         * instruction has not executed yet the breakpoint fires again for the SAME hit */
                        struct user_regs_struct r2; ptrace(PTRACE_GETREGS, t, 0, &r2);
                        if (r2.rip == addr) { refire_addr = addr; refire_rsp = r2.rsp; }
                        bp_arm(b); ptrace(PTRACE_CONT, t, 0, WSTOPSIG(st2)); continue;
                    }
                    bp_arm(b);
                }
                ptrace(PTRACE_CONT, t, 0, 0);
                continue;
            }
            ptrace(PTRACE_CONT, t, 0, 0);             /* exec/new-thread trap */
            continue;
        }
        if (sig == SIGSTOP) { ptrace(PTRACE_CONT, t, 0, 0); continue; }   /* new thread's initial stop */
        ptrace(PTRACE_CONT, t, 0, sig);               /* forward (Go uses SIGURG for preemption) */
    }
    fprintf(g_out, "\n ],\n \"exit_code\": %d}\n", exit_code);
    fflush(g_out);
    fprintf(stderr, "tracee exit code %d\n", exit_code);
    return exit_code == 0 ? 0 : 1;
}
