// hconv_bl.cpp — the reference's slot-packed baseline ("BL") convolution, the first half of its `conv k i n` run
// (main.go:639-640), on the MI355X engine. Next-row 8f-2 of the scope table.
//
// Reference mapping (file:line -> here):
//   test_BL.go:16-185 testConv_BL_in           -> testConv_BL_in   (boot = true: + blBootReLU of hconv_relu.cpp, test_BL.go:113-168)
//   main.go:101-112,413-435 newContext("BL_Conv") -> bl_newContext (params set [7], rotations {a*W+b} U {r*W^2},
//                                                   rotation keys over P = {0x1fffffffffe00001, 0x1fffffffffc80001})
//   eval.go:78-134   evalConv_BN_BL_test       -> evalConv_BN_BL_test
//   conv.go:120-143  preConv_BL                -> preConv_BL   (RotateHoisted: here one key switch per rotation; the
//                                                 hoisted and the plain rotation compute the same residues)
//   conv.go:146-178  postConv_BL               -> postConv_BL
//   conv.go:57-116   reshape_input_BL/reshape_ker_BL, main.go:1073-1103 post_trim_BL/post_process_BL
//   Lattigo ckks.encoderComplex128.Encode/Decode (special FFT over the rotation group 5^j) -> Encoder (host, fp64)
// Every residue operation is a C-ABI call on the GPU: hc_mul / hc_add per limb, hc_keyswitch (general hybrid key
// switch, level 1, two special primes) + hc_permute for rotations, hc_ntt for ToNTT. The slot encoder is host fp64
// code exactly as in the reference's timed region (conv.go:165-166); it is validated by decrypted precision, the
// integer pipeline by bit-exact parity with the oracle and the reference binary's key-switch digests.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <complex>
#include <map>
#include <random>
#include <set>

#include "hconv_host.hpp"
#include "hconv_encoder.hpp"

namespace hconv {

#define HCB(c, call) do { int rc_ = (call); if (rc_) panic(std::string(#call) + ": " + hc_last_error(c)); } while (0)

static const uint64_t BLQ[4] = {0x80000000080001ull, 0x10000000006e0001ull, 0x1fffffffffe00001ull, 0x1fffffffffc80001ull};   // Q0, Q1 of set [7]; P0, P1
static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t r = a + b; return r >= q ? r - q : r; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static inline uint64_t to_mont(uint64_t a, uint64_t q) { return (uint64_t)((((u128)a) << 64) % q); }
static std::string dur(std::chrono::steady_clock::time_point t0) {
    double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    char b[64];
    if (ns < 1e6) snprintf(b, sizeof b, "%.6gµs", ns / 1e3); else if (ns < 1e9) snprintf(b, sizeof b, "%.9gms", ns / 1e6); else snprintf(b, sizeof b, "%.9gs", ns / 1e9);
    return b;
}
static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }

struct BLContext {
    hc_ctx *hc = nullptr;
    int in_wid = 0;
    std::vector<int64_t> sk;
    std::vector<uint64_t> sk_ntt[4];
    ChaChaRng g;
    std::set<uint64_t> keys;
    double scale = (double)(1 << 30);
    Boot *btp = nullptr;          // cont.btp (main.go:476): the stock bootstrapper over parameter set [7], only for convReLU
    Seed256 seed;
};
struct BLCt { uint64_t *d = nullptr; double Scale = 0; };     // level-1 ciphertext: device [poly 2][limb 2][N]

static uint64_t *bl_rows(BLContext *c, size_t rows) { void *p = nullptr; HCB(c->hc, hc_malloc(c->hc, rows * N * 8, &p)); return (uint64_t *)p; }
static std::vector<uint64_t> bl_ntt(BLContext *c, int mod, std::vector<uint64_t> rows, bool inverse = false) {
    uint64_t *d = bl_rows(c, rows.size() / N); HCB(c->hc, hc_upload(c->hc, d, rows.data(), rows.size() * 8));
    HCB(c->hc, inverse ? hc_intt(c->hc, mod, d, d, (int)(rows.size() / N)) : hc_ntt(c->hc, mod, d, d, (int)(rows.size() / N)));
    HCB(c->hc, hc_download(c->hc, rows.data(), d, rows.size() * 8)); HCB(c->hc, hc_free(c->hc, d));
    return rows;
}
static std::vector<uint64_t> signed_row(const std::vector<int64_t> &v, uint64_t q) {
    std::vector<uint64_t> r(N);
    for (int j = 0; j < N; j++) r[(size_t)j] = v[(size_t)j] >= 0 ? (uint64_t)v[(size_t)j] % q : q - ((uint64_t)(-v[(size_t)j]) % q);
    return r;
}
static std::vector<int64_t> gaussian(BLContext *c) {
    std::vector<int64_t> e(N); std::normal_distribution<double> d(0.0, 3.2);
    for (auto &x : e) { double v; do { v = d(c->g); } while (fabs(v) > 19.2); x = (int64_t)llround(v); }
    return e;
}
static std::vector<uint64_t> uniform_row(BLContext *c, uint64_t q) { std::vector<uint64_t> r(N); std::uniform_int_distribution<uint64_t> d(0, q - 1); for (auto &x : r) x = d(c->g); return r; }

// params.GaloisElementForColumnRotationBy(k) = 5^(k mod 2N) mod 2N
static uint64_t gal_for_rotation(int k) {
    const uint64_t twoN = 2ull * N; uint64_t e = (uint64_t)(int64_t)k & (twoN - 1), r = 1, b = 5;
    while (e) { if (e & 1) r = (r * b) % twoN; b = (b * b) % twoN; e >>= 1; }
    return r;
}
// rlwe.GenRotationKeys restricted to what a level-1 key switch reads: one digit over {Q0,Q1}, P = {P0,P1}
static void bl_gen_key(BLContext *c, uint64_t galEl) {
    if (c->keys.count(galEl)) return;
    const uint64_t twoN = 2ull * N; uint64_t ginv = 1, b = galEl % twoN;
    for (uint64_t e = twoN - 1; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN;
    std::vector<int64_t> sko(N, 0);
    for (int i = 0; i < N; i++) { uint64_t t = ((uint64_t)i * ginv) % twoN; if (t < (uint64_t)N) sko[t] = c->sk[(size_t)i]; else sko[t - N] = -c->sk[(size_t)i]; }
    std::vector<int64_t> e = gaussian(c);
    std::vector<uint64_t> rows((size_t)2 * 4 * N);                // [k][limb Q0,Q1,P0,P1][N]
    for (int T = 0; T < 4; T++) {
        const uint64_t q = BLQ[T];
        std::vector<uint64_t> a = uniform_row(c, q), both = signed_row(sko, q), en = signed_row(e, q);
        both.insert(both.end(), en.begin(), en.end());
        both = bl_ntt(c, T, both);
        const uint64_t pmod = T < 2 ? mulmod(BLQ[2] % q, BLQ[3] % q, q) : 0;     // P*s on the Q limbs of the (single) digit
        for (int j = 0; j < N; j++) {
            uint64_t v = submod(both[(size_t)N + j], mulmod(a[(size_t)j], both[(size_t)j], q), q);
            v = addmod(v, mulmod(pmod, c->sk_ntt[T][(size_t)j], q), q);
            rows[((size_t)0 * 4 + T) * N + j] = to_mont(v, q); rows[((size_t)1 * 4 + T) * N + j] = to_mont(a[(size_t)j], q);
        }
    }
    HCB(c->hc, hc_swk_load(c->hc, galEl, 1, rows.data()));
    c->keys.insert(galEl);
}

// ---------------------------------------------------------------- level-1 evaluator ops on the C ABI
static BLCt bl_alloc(BLContext *c, double scale) { BLCt r; r.d = bl_rows(c, 4); r.Scale = scale; return r; }
static void bl_free(BLContext *c, BLCt &ct) { if (ct.d) HCB(c->hc, hc_free(c->hc, ct.d)); ct.d = nullptr; }
static BLCt MulNew(BLContext *c, const BLCt &ct, const uint64_t *pt, double pt_scale) {           // conv.go:168,170
    BLCt r = bl_alloc(c, ct.Scale * pt_scale);
    for (int p = 0; p < 2; p++) for (int l = 0; l < 2; l++) HCB(c->hc, hc_mul(c->hc, l, ct.d + ((size_t)p * 2 + l) * N, pt + (size_t)l * N, r.d + ((size_t)p * 2 + l) * N, 1));
    return r;
}
static void Add(BLContext *c, const BLCt &a, const BLCt &b, BLCt &out) {                             // conv.go:171, eval.go:123
    for (int p = 0; p < 2; p++) for (int l = 0; l < 2; l++) HCB(c->hc, hc_add(c->hc, l, a.d + ((size_t)p * 2 + l) * N, b.d + ((size_t)p * 2 + l) * N, out.d + ((size_t)p * 2 + l) * N, 1));
}
static BLCt RotateNew(BLContext *c, const BLCt &ct, int k) {       // evaluator.RotateNew -> permuteNTT: key switch c1, + c0, permute both
    const uint64_t gal = gal_for_rotation(k);
    BLCt r = bl_alloc(c, ct.Scale);
    if (gal == 1) { HCB(c->hc, hc_copy(c->hc, r.d, ct.d, (size_t)4 * N * 8)); return r; }
    uint64_t *d = bl_rows(c, 4);                                   // d0[2 limbs] | d1[2 limbs]
    HCB(c->hc, hc_keyswitch(c->hc, gal, 1, ct.d + (size_t)2 * N, d, d + (size_t)2 * N));
    for (int l = 0; l < 2; l++) HCB(c->hc, hc_add(c->hc, l, d + (size_t)l * N, ct.d + (size_t)l * N, d + (size_t)l * N, 1));
    HCB(c->hc, hc_permute(c->hc, gal, d, r.d, 4));
    HCB(c->hc, hc_free(c->hc, d));
    return r;
}

// ---------------------------------------------------------------- reference layout functions
static std::vector<cplx> reshape_input_BL(const std::vector<double> &input, int in_wid) {              // conv.go:57-72
    std::vector<cplx> out(input.size()); const int batch = (int)input.size() / (in_wid * in_wid); size_t l = 0;
    for (int i = 0; i < in_wid; i++) for (int j = 0; j < in_wid; j++) for (int k = 0; k < batch; k++) out[(size_t)(i * in_wid + j + k * in_wid * in_wid)] = cplx(input[l++], 0);
    return out;
}
// conv.go:78-116 (trans = false): max_ker_rs[i][j][ib][ob]
typedef std::vector<double> Ker4;   // flat [ker_wid][ker_wid][max_bat][max_bat]
static Ker4 reshape_ker_BL(const std::vector<double> &input, const std::vector<double> &BN_a, int ker_wid, int inB, int outB, int max_bat, int norm) {
    Ker4 out((size_t)ker_wid * ker_wid * max_bat * max_bat, 0.0);
    for (int i = 0; i < ker_wid; i++) for (int j = 0; j < ker_wid; j++) for (int ib = 0; ib < inB; ib++) for (int ob = 0; ob < outB; ob++)
        out[(((size_t)i * ker_wid + j) * max_bat + norm * ib) * max_bat + norm * ob] = input[(size_t)(ob + ib * outB + j * outB * inB + i * outB * inB * ker_wid)] * BN_a[(size_t)ob];
    return out;
}
static std::vector<double> post_trim_BL(const std::vector<cplx> &in_vals, int raw_in_wid, int in_wid) {   // main.go:1073-1086
    const int batch = (int)in_vals.size() / (in_wid * in_wid); std::vector<double> out((size_t)raw_in_wid * raw_in_wid * batch);
    for (int b = 0; b < batch; b++) for (int i = 0; i < raw_in_wid; i++) for (int j = 0; j < raw_in_wid; j++)
        out[(size_t)(b * raw_in_wid * raw_in_wid + i * raw_in_wid + j)] = in_vals[(size_t)(b * in_wid * in_wid + i * in_wid + j)].real();
    return out;
}
static std::vector<double> post_process_BL(const std::vector<double> &in_vals, int raw_in_wid) {        // main.go:1089-1103
    const int batch = (int)in_vals.size() / (raw_in_wid * raw_in_wid); std::vector<double> out(in_vals.size());
    for (int i = 0; i < raw_in_wid; i++) for (int j = 0; j < raw_in_wid; j++) for (int b = 0; b < batch; b++)
        out[(size_t)(i * raw_in_wid * batch + batch * j + b)] = in_vals[(size_t)(b * raw_in_wid * raw_in_wid + i * raw_in_wid + j)];
    return out;
}

// ---------------------------------------------------------------- conv.go:120-178, eval.go:78-134
// conv.go:120-143: evaluator.RotateHoisted of the input by every kernel offset: one digit decomposition of c1 for all of them
static std::vector<BLCt> preConv_BL(BLContext *c, const BLCt &ct_in, int in_wid, int ker_wid) {
    std::vector<BLCt> rots; const int st = -(ker_wid / 2), end = ker_wid / 2;
    HCB(c->hc, hc_keyswitch_decompose(c->hc, 1, ct_in.d + (size_t)2 * N));
    uint64_t *d = bl_rows(c, 4);
    for (int i = st; i <= end; i++) for (int j = st; j <= end; j++) {
        const uint64_t gal = gal_for_rotation(i * in_wid + j);
        BLCt r = bl_alloc(c, ct_in.Scale);
        if (gal == 1) { HCB(c->hc, hc_copy(c->hc, r.d, ct_in.d, (size_t)4 * N * 8)); rots.push_back(r); continue; }
        HCB(c->hc, hc_keyswitch_hoisted(c->hc, gal, 1, ct_in.d + (size_t)2 * N, d, d + (size_t)2 * N));
        for (int l = 0; l < 2; l++) HCB(c->hc, hc_add(c->hc, l, d + (size_t)l * N, ct_in.d + (size_t)l * N, d + (size_t)l * N, 1));
        HCB(c->hc, hc_permute(c->hc, gal, d, r.d, 4));
        rots.push_back(r);
    }
    HCB(c->hc, hc_free(c->hc, d));
    return rots;
}
// conv.go:146-178 for one output rotation. The k^2 plaintexts of a rotation are built ON THE DEVICE: hc_bl_post_ker_slots writes the slot
// vectors postKer (conv.go:150-164) from the device copy of max_ker_rs, hc_encode_slots runs Lattigo's encoder (special inverse FFT in
// fp64, scaleUpVecExact, ToNTT: conv.go:165-166) on all of them at once -- the same residues the host encoder produces (both are compared
// with the oracle that is pinned to the reference binary's own Encode digests).
static BLCt postConv_BL(BLContext *c, const std::vector<BLCt> &ct_in_rots, int in_wid, int ker_wid, int rot, int pad, const double *d_max_ker_rs, int max_batch, double *d_slots, uint64_t *d_pl) {
    const int taps = ker_wid * ker_wid;
    HCB(c->hc, hc_bl_post_ker_slots(c->hc, d_max_ker_rs, in_wid, ker_wid, pad, max_batch, rot, d_slots));
    HCB(c->hc, hc_encode_slots(c->hc, d_slots, taps, 1, c->scale, 1, d_pl));                              // conv.go:165-166 for every tap
    BLCt ct_out = bl_alloc(c, ct_in_rots[0].Scale * c->scale);                                             // conv.go:167-172: MulNew per tap, Add
    std::vector<const uint64_t *> cts((size_t)taps);
    for (int it = 0; it < taps; it++) cts[(size_t)it] = ct_in_rots[(size_t)it].d;
    HCB(c->hc, hc_lv_mul_sum(c->hc, 1, cts.data(), d_pl, taps, ct_out.d));
    return ct_out;
}
static BLCt evalConv_BN_BL_test(BLContext *c, const Encoder &enc, const BLCt &ct_input, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                                const std::vector<double> &bn_b, int in_wid, int ker_wid, int real_ib, int real_ob, int pos, int norm, int pad) {
    (void)pos;
    const int in_size = in_wid * in_wid, out_size = in_size, max_batch = N / (2 * in_size);
    auto start = now();
    Ker4 max_ker_rs = reshape_ker_BL(ker_in, bn_a, ker_wid, real_ib, real_ob, max_batch, norm);
    const double scale_exp = c->scale * c->scale;
    std::vector<cplx> bn_b_slots((size_t)N / 2, cplx(0, 0));
    for (size_t i = 0; i < bn_b.size(); i++) for (int j = 0; j < in_wid - pad; j++) for (int k = 0; k < in_wid - pad; k++)
        bn_b_slots[(size_t)(j + k * in_wid + norm * out_size * (int)i)] = cplx(bn_b[i], 0);             // eval.go:93-99
    uint64_t *pl_bn_b = bl_rows(c, 2);
    const int taps = ker_wid * ker_wid;
    void *vp = nullptr; HCB(c->hc, hc_malloc(c->hc, (size_t)taps * (N / 2) * 16, &vp)); double *d_slots = (double *)vp;     // [taps][N/2] complex128
    HCB(c->hc, hc_malloc(c->hc, max_ker_rs.size() * sizeof(double), &vp)); double *d_ker = (double *)vp;
    HCB(c->hc, hc_upload(c->hc, d_ker, max_ker_rs.data(), max_ker_rs.size() * sizeof(double)));
    uint64_t *d_pl = bl_rows(c, (size_t)taps * 2);
    HCB(c->hc, hc_upload(c->hc, d_slots, bn_b_slots.data(), bn_b_slots.size() * 16));
    HCB(c->hc, hc_encode_slots(c->hc, d_slots, 1, 1, scale_exp, 1, pl_bn_b));                                          // eval.go:101-102 EncodeNTT
    printf("Plaintext (kernel) preparation, Done in %s \n", dur(start).c_str());
    start = now();
    std::vector<BLCt> ct_inputs_rots = preConv_BL(c, ct_input, in_wid, ker_wid);
    HCB(c->hc, hc_sync(c->hc));
    printf("preConv done in %s \n", dur(start).c_str());
    const int rot_iters = (norm * real_ob == max_batch) ? real_ob : max_batch;
    BLCt ct_res;
    for (int i = 0; i < rot_iters; i++) {
        BLCt ct_tmp = postConv_BL(c, ct_inputs_rots, in_wid, ker_wid, norm * i, pad, d_ker, max_batch, d_slots, d_pl);
        if (i == 0) ct_res = ct_tmp;
        else { BLCt r = RotateNew(c, ct_tmp, norm * i * out_size); Add(c, ct_res, r, ct_res); bl_free(c, r); bl_free(c, ct_tmp); }   // eval.go:123
    }
    if (ct_res.Scale != scale_exp) panic("Different scale between pl_bn_b and ctxt");                    // eval.go:127-129
    for (int l = 0; l < 2; l++) HCB(c->hc, hc_add(c->hc, l, ct_res.d + (size_t)l * N, pl_bn_b + (size_t)l * N, ct_res.d + (size_t)l * N, 1));   // eval.go:130
    HCB(c->hc, hc_sync(c->hc));
    printf("Conv (with BN) Done in %s \n", dur(start).c_str());
    for (auto &r : ct_inputs_rots) bl_free(c, r);
    HCB(c->hc, hc_free(c->hc, pl_bn_b)); HCB(c->hc, hc_free(c->hc, d_slots)); HCB(c->hc, hc_free(c->hc, d_ker)); HCB(c->hc, hc_free(c->hc, d_pl));
    (void)enc;
    return ct_res;
}

// ---------------------------------------------------------------- context, encrypt, decrypt
static BLContext *bl_newContext(int logN, int ker_wid, int in_wid, bool boot) {
    auto cont_start = now();
    BLContext *c = new BLContext(); c->in_wid = in_wid;
    // ckks.DefaultBootstrapParams[7] has 28 Q primes + 5 P primes, logQP = 1582 (the figure the reference prints); this path
    // only ever touches Q0, Q1 and, for a level-1 key switch, the first two special primes
    printf("CKKS parameters: logN = %d, logSlots = %d, h = %d, logQP = %d, levels = %d, scale= 2^%f, sigma = %f \n", LOGN, LOGN - 1, 192, 1582, 28, 30.0, 3.2);
    if ((1 << logN) != N) { printf("Set Boot logN to %d\n", logN); panic("Boot N != N"); }
    c->seed = seedFromEnvironment(); c->g.reseed(c->seed, 0x424c);
    const int dev = getenv("HCONV_DEVICE") ? atoi(getenv("HCONV_DEVICE")) : 0;
    if (hc_ctx_create(&c->hc, LOGN, BLQ, 2, BLQ + 2, 2, dev)) panic(std::string("hc_ctx_create: ") + hc_last_error(nullptr));
    applyEnvOptions(c->hc);
    c->sk.assign(N, 0);
    { int placed = 0; while (placed < 192) { uint64_t r = c->g(); int pos = (int)(r % N); if (!c->sk[(size_t)pos]) { c->sk[(size_t)pos] = (r >> 40) & 1 ? 1 : -1; placed++; } } }
    for (int m = 0; m < 4; m++) c->sk_ntt[m] = bl_ntt(c, m, signed_row(c->sk, BLQ[m]));
    // main.go:101-112: rotations {k*W + k2} and {r * W^2}; removeDuplicateInt
    std::vector<int> rotations; std::set<int> seen;
    auto push = [&](int r) { if (!seen.count(r)) { seen.insert(r); rotations.push_back(r); } };
    for (int k = -(ker_wid / 2); k <= ker_wid / 2; k++) for (int k2 = -(ker_wid / 2); k2 <= ker_wid / 2; k2++) push(k * in_wid + k2);
    const int out_batch = (N / 2) / (in_wid * in_wid);
    for (int k = 1; k < out_batch; k++) push(k * in_wid * in_wid);
    printf("Num Rotations:  %d\n", (int)rotations.size());
    for (int r : rotations) bl_gen_key(c, gal_for_rotation(r));
    if (boot) {                                                            // main.go:464-507
        printf("Generating bootstrapping keys...\n");
        c->btp = newBootBL(c->sk, c->seed, dev);
        printf("Done in %s \n", dur(cont_start).c_str());
    }
    return c;
}
static BLCt bl_encrypt(BLContext *c, const std::vector<uint64_t> &pt_rows, double scale) {
    std::vector<int64_t> e = gaussian(c); std::vector<uint64_t> ct((size_t)4 * N);
    for (int l = 0; l < 2; l++) {
        const uint64_t q = BLQ[l]; std::vector<uint64_t> c1 = uniform_row(c, q), t = signed_row(e, q);
        for (int j = 0; j < N; j++) t[(size_t)j] = addmod(t[(size_t)j], pt_rows[(size_t)l * N + j] % q, q);
        t = bl_ntt(c, l, t);
        for (int j = 0; j < N; j++) { ct[(size_t)l * N + j] = submod(t[(size_t)j], mulmod(c1[(size_t)j], c->sk_ntt[l][(size_t)j], q), q); ct[(size_t)(2 + l) * N + j] = c1[(size_t)j]; }
    }
    BLCt r = bl_alloc(c, scale); HCB(c->hc, hc_upload(c->hc, r.d, ct.data(), ct.size() * 8));
    return r;
}
// Decrypt at level 1 + encoder.Decode: CRT over Q0*Q1 (115 bits), centre, /scale, special FFT
static std::vector<cplx> bl_decrypt_decode(BLContext *c, const Encoder &enc, const BLCt &ct) {
    std::vector<uint64_t> h((size_t)4 * N); HCB(c->hc, hc_download(c->hc, h.data(), ct.d, h.size() * 8));
    std::vector<uint64_t> m[2];
    for (int l = 0; l < 2; l++) {
        const uint64_t q = BLQ[l]; m[l].resize(N);
        for (int j = 0; j < N; j++) m[l][(size_t)j] = addmod(h[(size_t)l * N + j], mulmod(h[(size_t)(2 + l) * N + j], c->sk_ntt[l][(size_t)j], q), q);
        m[l] = bl_ntt(c, l, m[l], true);
    }
    const uint64_t q0 = BLQ[0], q1 = BLQ[1]; uint64_t inv = 1; { uint64_t b = q0 % q1, e = q1 - 2; while (e) { if (e & 1) inv = mulmod(inv, b, q1); b = mulmod(b, b, q1); e >>= 1; } }
    const u128 Q = (u128)q0 * q1;
    std::vector<cplx> vals((size_t)N / 2);
    std::vector<double> cf(N);
    for (int j = 0; j < N; j++) {
        const uint64_t a0 = m[0][(size_t)j], a1 = m[1][(size_t)j];
        const uint64_t t = mulmod(submod(a1 % q1, a0 % q1, q1), inv, q1);
        const u128 x = (u128)a0 + (u128)q0 * t;
        cf[(size_t)j] = x > Q / 2 ? -(double)(Q - x) / ct.Scale : (double)x / ct.Scale;
    }
    for (int i = 0; i < N / 2; i++) vals[(size_t)i] = cplx(cf[(size_t)i], cf[(size_t)(i + N / 2)]);
    enc.fft(vals);
    return vals;
}

// ---------------------------------------------------------------- test_BL.go:16-185 (boot = false)
void testConv_BL_in(int real_batch, int in_wid, int ker_wid, int total_test_num, bool boot) {
    const std::string test_dir = "test_conv_data/";
    const int pad = ker_wid / 2, raw_in_wid = in_wid - pad, in_size = in_wid * in_wid, ker_size = ker_wid * ker_wid;
    const int slots = real_batch / 2 * in_size; int log_slots = 0; while ((1 << log_slots) < slots) log_slots++;
    const int out_batch = real_batch, in_batch = real_batch;
    BLContext *cont = bl_newContext(log_slots + 1, ker_wid, in_wid, boot);
    Encoder enc;
    printf("vec size: log2 =  %d\n", LOGN);
    printf("raw input width:  %d\n", raw_in_wid);
    printf("kernel width:  %d\n", ker_wid);
    printf("num batches in & out:  %d ,  %d\n", real_batch, out_batch);
    for (int test_iter = 0; test_iter < total_test_num; test_iter++) {
        printf("%d -th iter...start\n", test_iter + 1);
        const std::string pre = test_dir + "test_conv" + std::to_string(ker_wid) + "_batch_" + std::to_string(in_batch) + "_", suf = "_" + std::to_string(test_iter) + ".csv";
        std::vector<double> input = readTxt(pre + "in" + suf, raw_in_wid * raw_in_wid * in_batch);
        std::vector<double> ker_in = readTxt(pre + "ker" + suf, in_batch * in_batch * ker_size);
        std::vector<double> bn_a = readTxt(pre + "bna" + suf, in_batch), bn_b = readTxt(pre + "bnb" + suf, in_batch);
        std::vector<double> real_out = readTxt(pre + (boot ? "reluout" : "out") + suf, raw_in_wid * raw_in_wid * in_batch);   // test_BL.go:53-57
        const int hb = real_batch / 2;
        std::vector<double> pad_input1((size_t)in_size * hb, 0.0), pad_input2((size_t)in_size * hb, 0.0);
        for (int i = 0; i < raw_in_wid; i++) for (int j = 0; j < raw_in_wid; j++) for (int b = 0; b < hb; b++) {
            pad_input1[(size_t)(b + j * hb + i * hb * in_wid)] = input[(size_t)(b + j * real_batch + i * real_batch * raw_in_wid)];
            pad_input2[(size_t)(b + j * hb + i * hb * in_wid)] = input[(size_t)(b + hb + j * real_batch + i * real_batch * raw_in_wid)];
        }
        std::vector<double> bn_a_sep[2], bn_b_sep[2], zeros((size_t)hb, 0.0);
        for (int out = 0; out < 2; out++) for (int i = 0; i < hb; i++) { bn_a_sep[out].push_back(bn_a[(size_t)(i + out * hb)]); bn_b_sep[out].push_back(bn_b[(size_t)(i + out * hb)]); }
        std::vector<double> ker_in_sep[2][2];
        for (int out = 0; out < 2; out++) for (int in = 0; in < 2; in++) {
            ker_in_sep[out][in].resize(ker_in.size() / 4);
            for (int k = 0; k < ker_size; k++) for (int i = 0; i < hb; i++) for (int j = 0; j < hb; j++)
                ker_in_sep[out][in][(size_t)(k * hb * hb + i * hb + j)] = ker_in[(size_t)(k * real_batch * real_batch + (i + in * hb) * real_batch + out * hb + j)];
        }
        auto start = now();
        BLCt ct_input1 = bl_encrypt(cont, enc.Encode(reshape_input_BL(pad_input1, in_wid), cont->scale, BLQ, 2), cont->scale);
        BLCt ct_input2 = bl_encrypt(cont, enc.Encode(reshape_input_BL(pad_input2, in_wid), cont->scale, BLQ, 2), cont->scale);
        printf("Encryption done in %s \n", dur(start).c_str());
        auto start_eval = now();
        BLCt ct_res[2];
        for (int pos = 0; pos < 2; pos++) {
            BLCt a = evalConv_BN_BL_test(cont, enc, ct_input1, ker_in_sep[pos][0], bn_a_sep[pos], bn_b_sep[pos], in_wid, ker_wid, hb, hb, 0, 1, pad);
            BLCt b = evalConv_BN_BL_test(cont, enc, ct_input2, ker_in_sep[pos][1], bn_a_sep[pos], zeros, in_wid, ker_wid, hb, hb, 0, 1, pad);
            Add(cont, a, b, a); bl_free(cont, b); ct_res[pos] = a;                                        // test_BL.go:108 AddNew
        }
        HCB(cont->hc, hc_sync(cont->hc));
        printf("Evaluation total done in %s \n", dur(start_eval).c_str());
        if (boot) {                                                                                       // test_BL.go:113-168
            const double alpha = 0.0, pow_ = 4.0;
            BLCt o0 = bl_alloc(cont, 0), o1 = bl_alloc(cont, 0); double sc = 0;
            blBootReLU(cont->btp, ct_res[0].d, ct_res[1].d, ct_res[0].Scale, alpha, pow_, o0.d, o1.d, &sc);
            bl_free(cont, ct_res[0]); bl_free(cont, ct_res[1]);
            o0.Scale = o1.Scale = sc; ct_res[0] = o0; ct_res[1] = o1;
        }
        start = now();
        std::vector<cplx> vals_tmp1 = bl_decrypt_decode(cont, enc, ct_res[0]), vals_tmp2 = bl_decrypt_decode(cont, enc, ct_res[1]);
        printf("Decryption Done in %s \n", dur(start).c_str());
        std::vector<double> test_out = post_trim_BL(vals_tmp1, raw_in_wid, in_wid), t2 = post_trim_BL(vals_tmp2, raw_in_wid, in_wid);
        test_out.insert(test_out.end(), t2.begin(), t2.end());
        test_out = post_process_BL(test_out, raw_in_wid);
        printDebugCfsPlain(test_out, real_out);
        bl_free(cont, ct_input1); bl_free(cont, ct_input2); bl_free(cont, ct_res[0]); bl_free(cont, ct_res[1]);
    }
    if (cont->btp) freeBoot(cont->btp);
    hc_ctx_destroy(cont->hc); delete cont;
}

}  // namespace hconv
