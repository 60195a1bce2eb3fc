// hconv_relu.cpp — the `convReLU` chain (scope row 8f-1) on the MI355X engine: everything after the convolution in
// eval.go:272-607 evalConv_BNRelu_new for kind "Conv" (log_sparse 0, iter 2), with device-resident ciphertexts.
//
// Reference mapping (file:line -> here):
//   eval.go:433-437  evalConv_BN at out_scale 2^(round(log2 Q0) - (pow+8)); Scale *= 2^pow      -> evalConv_BNRelu_new
//   eval.go:450      cont.btp.BootstrappConv_CtoS (fork-only: test_run ckks.(*Bootstrapper).BootstrappConv_CtoS:
//                    modUp, CoeffsToSlots, evaluateSine)                                         -> Boot::ctos
//   conv.go:435-480  evalReLU (EvaluatePoly x3, AddConst, Mul, Relinearize)                      -> evalReLU
//   eval.go:474      MulByPow2                                                                    -> mul_const_int(2^pow)
//   conv.go:417-431  keep_ctxt, rot_util.go:141-174 gen_keep_vec                                 -> keep_ctxt, gen_keep_vec
//   eval.go:550      cont.btp.BootstrappConv_StoC                                                 -> Boot::stoc
//   main.go:464-507  bootstrapping keys (rlk + rotation keys over the five special primes)       -> Boot::key (generated on demand)
// The bootstrapper exists only inside the un-vendored Lattigo fork; this is a restatement on the same modulus chain and level
// assignment (parameter set [6]): levels 27..24 CoeffsToSlots, 23..16 sine (Chebyshev degree 63 of the cosine,
// two double angles, K = 25, message ratio 256), 15..5 the ReLU polynomials, 5 the mask, 3..2 SlotsToCoeffs. It mirrors
// tests/oracle_ckks.py statement by statement; that file run on the oracle and on this library's C ABI gives bit-identical
// ciphertexts at every stage (tests/test_gpu_a_parity.py::test_conv_relu_tail_on_gpu). All residue arithmetic is C-ABI calls:
// hc_lv_* (all limbs per launch), hc_keyswitch (hybrid, alpha = 5), hc_div_round_last, hc_permute; the slot encoder and the
// float64 scale bookkeeping are host code, as in the reference.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <set>

#include "hconv_encoder.hpp"
#include "hconv_sha256.hpp"
#include "hconv_sine_coeffs.hpp"
#include "hconv_host.hpp"

namespace hconv {

#define HCR(call) do { int rc_ = (call); if (rc_) panic(std::string(#call) + ": " + hc_last_error(hc)); } while (0)
static const int LV_CTS_TOP = 27, LV_SINE_TOP = 23, LV_RELU_TOP = 15;     // the same in parameter sets [6] and [7]: CtS 27..24, sine 23..16
static const int SIN_K = 25, SIN_DEG = 63, SIN_DOUBLE = 2;
typedef std::map<int, std::vector<cplx>> DiagMat;        // rotation k -> diagonal (n complex values)

static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static std::string dur(std::chrono::steady_clock::time_point t0) {
    double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    char b[64];
    if (ns < 1e6) snprintf(b, sizeof b, "%.6gµs", ns / 1e3); else if (ns < 1e9) snprintf(b, sizeof b, "%.9gms", ns / 1e6); else snprintf(b, sizeof b, "%.9gs", ns / 1e9);
    return b;
}
static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }

// device-resident ciphertext: deg+1 polynomials, each a pooled block of NQ rows of which rows 0..level are live
struct DCt {
    std::shared_ptr<uint64_t> p[3];
    int deg = 1, level = 0;
    double scale = 0;
};
struct DPt { std::shared_ptr<uint64_t> p; int level = 0; double scale = 0; };

struct Boot;
static void profile_dump(Boot *B, const char *label);      // HCONV_PROFILE=1: per-kernel HIP-event totals since the last dump
static bool profiling() { static const bool on = getenv("HCONV_PROFILE") && atoi(getenv("HCONV_PROFILE")); return on; }

struct Boot {
    hc_ctx *hc = nullptr;
    std::vector<uint64_t> Q, P;
    int NQ = 0;
    // where SlotsToCoeffs sits and what its plaintext scales are: set [6] (Ours, main.go:52): levels 3..2 after the ReLU, two matrices
    // at sqrt(q3) and one at 2^30 (input scale 2^60 -> output 2^30 at level 1); set [7] (BL, main.go:54: the stock Bootstrapp):
    // levels 15..14 right after the sine, 2^40 each (input 2^30 -> output 2^30 * 2^120 / (q15 q14) at level 13)
    int chain = 6, LV_STC_TOP = 3, lv_relin_lo = LV_RELU_TOP - 11;
    double stc_scale_top = 0, stc_scale_last = 1073741824.0;
    // scale the sine leaves its result at (level 15). Ours: 2^30, what evalReLU consumes. Baseline: 2^55 — its SlotsToCoeffs follows
    // directly and multiplies whatever noise its input carries by ~sqrt(N) relative to the slot values, so the input must sit far above
    // the key-switch noise (the reference's evaluateSine likewise hands over at 2^55..2^60: gotrace -noplant log of `convReLU 3 0 1`)
    double sine_out_scale = 1073741824.0;
    const int n = N / 2;
    std::vector<int64_t> sk;
    uint64_t *d_sk = nullptr;                               // [NQ+NP][N] NTT rows of sk on the device
    std::map<std::pair<uint64_t, int>, uint64_t> key_ids;   // (galEl or 0 = relinearisation, level) -> id loaded with hc_swk_load
    std::vector<uint64_t *> pool;
    ChaChaRng rng;                    // the bootstrapper's own key stream (switching keys, encryption masks)
    Encoder enc;
    std::shared_ptr<uint64_t> mono_i;                        // NTT(X^(N/2)) for every limb
    struct LT { int n1 = 1; std::map<int, std::map<int, DPt>> giant; double pt_scale = 0; int level = 0; bool qp = false; };   // the diagonals are encoded mod Q_0..Q_level AND mod every P (rows [level+1+np][N]) for linear_transform_qp
    struct Set { int ls = 0, ns = 0; std::vector<LT> cts, stc; };      // one bootstrapper of the reference (btp, btp2..btp5: main.go:480-500)
    std::map<int, Set> sets;                                           // by log_sparse
    std::map<int, Encoder> sub_enc;                                    // encoders of the rings with fewer slots (sparse embedding), by log2 of their degree
    std::vector<double> sine;
    long n_keyswitch = 0, n_keys = 0;
    bool parts_merged = false;                               // ctos_fork returned ONE ciphertext of 2 nb images (both halves): see merge2
    // algorithmic traffic of what has been evaluated, in rows of N residues (SURVEY.md 8(d)'s convention carried to the chain: every evaluator operation reads its
    // ciphertext operands once and writes its result once, temporaries stay on chip; switching keys, diagonals and masks are read once per operation and - being common
    // to the images of a batch - once per launch set): alg_ct per ciphertext, alg_shared per launch set
    double alg_ct = 0, alg_shared = 0;
    int ks_rows(int L) const { const int a = (int)P.size(), nl = L + 1, nt = nl + a; return 2 * ((nl + a - 1) / a) * nt; }      // rows of one switching key at level L
    std::map<std::string, DPt> pt_cache;                     // encoded 0/1 masks (keep_ctxt, ext_double_ctxt), by what defines them

    // ---------------- memory and the image batch
    // HCONV_IMAGE_BATCH = nb_max > 1: every ciphertext object below stands for the ciphertexts of nb <= nb_max images going through the layer together (test.go:128 runs
    // them one after another): a polynomial block holds nb_max x NQ rows, image z at z * NQ rows, an extended-basis block nb_max x 2 (NQ + NP) rows; hc_set_batch makes
    // every leveled ABI call cover the nb images in one launch set (plaintexts - masks, diagonals, constants - and switching keys are shared and read once per launch).
    int nb_max = 1, nb = 1;
    size_t poly_stride() const { return (size_t)NQ * N; }
    size_t qp_stride() const { return (size_t)2 * (NQ + P.size()) * N; }
    void set_nb(int n) { if (n < 1 || n > nb_max) panic("image batch out of range"); nb = n; HCR(hc_set_batch(hc, n, poly_stride(), qp_stride())); }
    // 4-byte rows (include/hconv.h, option pack32 = 2: the bootstrapping contexts run with it unless HCONV_PACK32 says otherwise): the rows of the ~30-bit limbs of every leveled
    // operand on the device are N 4-byte words at the row's address. The host converts only where IT reads or writes such rows: rows[0 .. nl) <-> limbs 0 .. nl - 1 (rows beyond
    // are special primes: large). Ciphertexts enter and leave the chain at levels 0 / 1 (large limbs), so the layers themselves convert nothing.
    std::vector<char> row32;                                 // per limb: hc_row_is32
    void pack_rows(std::vector<uint64_t> &rows, int nl) const {
        for (int l = 0; l < nl && l < (int)row32.size(); l++) if (row32[(size_t)l]) { uint32_t *d = reinterpret_cast<uint32_t *>(&rows[(size_t)l * N]); for (int j = 0; j < N; j++) d[j] = (uint32_t)rows[(size_t)l * N + (size_t)j]; }
    }
    void unpack_rows(std::vector<uint64_t> &rows, int nl) const {
        for (int l = 0; l < nl && l < (int)row32.size(); l++) if (row32[(size_t)l]) { const uint32_t *d = reinterpret_cast<const uint32_t *>(&rows[(size_t)l * N]); for (int j = N - 1; j >= 0; j--) rows[(size_t)l * N + (size_t)j] = d[j]; }
    }
    struct Single {            // scope in which the ABI acts on ONE polynomial per call (encoding plaintexts, debugging)
        Boot *b; int saved;
        explicit Single(Boot *b_) : b(b_), saved(b_->nb) { if (saved != 1) b->set_nb(1); }
        ~Single() { if (saved != 1) b->set_nb(saved); }
    };
    struct Batch {             // scope in which every leveled ABI call covers n images; the context is back at ONE image per call on every way out of the scope
        Boot *b;
        Batch(Boot *b_, int n) : b(b_) { b->set_nb(n); }
        ~Batch() { b->nb = 1; hc_set_batch(b->hc, 1, 0, 0); }
        Batch(const Batch &) = delete; Batch &operator=(const Batch &) = delete;
    };
    std::shared_ptr<uint64_t> block() {
        uint64_t *d;
        if (!pool.empty()) { d = pool.back(); pool.pop_back(); }
        else { void *v = nullptr; HCR(hc_malloc(hc, (size_t)nb_max * NQ * N * 8, &v)); d = (uint64_t *)v; }
        return std::shared_ptr<uint64_t>(d, [this](uint64_t *x) { pool.push_back(x); });
    }
    std::vector<uint64_t *> pool1;
    std::shared_ptr<uint64_t> block1() {               // one polynomial whatever the batch: plaintexts
        uint64_t *d;
        if (!pool1.empty()) { d = pool1.back(); pool1.pop_back(); }
        else { void *v = nullptr; HCR(hc_malloc(hc, (size_t)NQ * N * 8, &v)); d = (uint64_t *)v; }
        return std::shared_ptr<uint64_t>(d, [this](uint64_t *x) { pool1.push_back(x); });
    }
    // two polynomials in the extended basis, [2][level+1+np][N] at the start of a 2 (NQ+NP)-row allocation (hc_keyswitch_qp / hc_mod_down2 layout), per image
    std::vector<uint64_t *> pool_qp;
    std::shared_ptr<uint64_t> block_qp2() {
        uint64_t *d;
        if (!pool_qp.empty()) { d = pool_qp.back(); pool_qp.pop_back(); }
        else { void *v = nullptr; HCR(hc_malloc(hc, (size_t)nb_max * qp_stride() * 8, &v)); d = (uint64_t *)v; }
        return std::shared_ptr<uint64_t>(d, [this](uint64_t *x) { pool_qp.push_back(x); });
    }
    // Full slots carry the two coefficient halves of every image as TWO ciphertexts through the sine and the ReLU (eval.go:462-477 loops over them). They see the same
    // operations at the same levels, so with room for 2 nb images per block (merge_parts) they ride as ONE batch of 2 nb "images": image z of the second half sits at
    // slot nb + z. The launches of the most expensive stages are then twice as wide for the same count - what HCONV_IMAGE_BATCH does across images, across the halves.
    bool merge_parts = false;
    DCt merge2(const DCt &a, const DCt &b) {              // called with nb = n: returns the 2n-image ciphertext and switches the batch to 2n
        if (a.level != b.level || a.deg != 1 || b.deg != 1 || 2 * nb > nb_max) panic("merge2: halves differ or the blocks are too small");
        DCt r = new_ct(a.level, 1, a.scale); const int n0 = nb;
        for (int d = 0; d < 2; d++) for (int z = 0; z < n0; z++) {
            HCR(hc_copy(hc, r.p[d].get() + (size_t)z * poly_stride(), a.p[d].get() + (size_t)z * poly_stride(), (size_t)(a.level + 1) * N * 8));
            HCR(hc_copy(hc, r.p[d].get() + (size_t)(n0 + z) * poly_stride(), b.p[d].get() + (size_t)z * poly_stride(), (size_t)(a.level + 1) * N * 8));
        }
        set_nb(2 * n0); alg_ct_at_merge = alg_ct;
        return r;
    }
    double alg_ct_at_merge = 0;                            // the merged stretch works on TWO ciphertexts per image: its per-ciphertext byte count is doubled at split2
    void split2(const DCt &m, DCt out[2]) {               // called with nb = 2n: the two halves as n-image ciphertexts; the batch goes back to n
        const int n0 = nb / 2; set_nb(n0); alg_ct += alg_ct - alg_ct_at_merge;
        for (int h = 0; h < 2; h++) {
            out[h] = new_ct(m.level, 1, m.scale);
            for (int d = 0; d < 2; d++) for (int z = 0; z < n0; z++)
                HCR(hc_copy(hc, out[h].p[d].get() + (size_t)z * poly_stride(), m.p[d].get() + (size_t)(h * n0 + z) * poly_stride(), (size_t)(m.level + 1) * N * 8));
        }
    }
    // rows [0, rows) of every image's polynomial: device-to-device copies between batched blocks
    void copy_rows(uint64_t *dst, const uint64_t *src, size_t rows, size_t dst_off_rows = 0, size_t src_off_rows = 0) {
        for (int z = 0; z < nb; z++) HCR(hc_copy(hc, dst + (size_t)z * poly_stride() + dst_off_rows * N, src + (size_t)z * poly_stride() + src_off_rows * N, rows * N * 8));
    }
    DCt new_ct(int level, int deg, double scale) { DCt c; c.deg = deg; c.level = level; c.scale = scale; for (int i = 0; i <= deg; i++) c.p[i] = block(); return c; }
    static DCt drop_to(const DCt &a, int level) { if (level > a.level) panic("drop_to: level above the ciphertext's"); DCt c = a; c.level = level; return c; }

    // ---------------- sampling (harness only; the reference's randomness is crypto/rand and unseeded): this stream keys the device's key generator
    uint64_t next() { return rng(); }
    uint64_t modulus(int T, int nl) const { return T < nl ? Q[(size_t)T] : P[(size_t)(T - nl)]; }
    int modidx(int T, int nl) const { return T < nl ? T : NQ + (T - nl); }

    // ---------------- keys (rlwe.GenSwitchingKey restricted to the limbs a level-`level` key switch reads)
    // kind: which key switch reads the key (0: SwitchKeysInPlace - relinearisation, conjugation, plain rotations; 1: the hoisted baby steps of a
    // linear transform; 2: its giant steps). Real keys do not depend on it. HCONV_CHAIN_REPLAY=<seed> (tests only): no key is generated;
    // every key holds the rows `gotrace -chain` planted into the reference binary for its kind (oracle/pin/gotrace.c: SEED_KSEVK with id 40 for
    // the baby steps, 41 for everything else), so that the chain can be compared with the binary's digests (tests/test_gpu_z_cli.py).
    uint64_t replay_seed = 0;
    std::vector<uint64_t> replay_rows; int replay_rows_id = -1, replay_rows_level = -1;
    static uint64_t splitmix_at(uint64_t seed, uint64_t i) { uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    uint64_t key(uint64_t gal, int level, int kind = 0) {
        const uint64_t ident = replay_seed ? (gal | ((uint64_t)(kind == 1 ? 1 : 2) << 40)) : gal;
        auto it = key_ids.find({ident, level});
        if (it != key_ids.end()) return it->second;
        const int alpha = (int)P.size(), nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha;
        if (replay_seed) {
            const int pid = kind == 1 ? 40 : 41;
            if (replay_rows_id != pid || replay_rows_level != level) {
                replay_rows.resize((size_t)beta * 2 * nt * N);
                for (int d = 0; d < beta; d++) for (int k = 0; k < 2; k++) for (int T = 0; T < nt; T++) {
                    const uint64_t q = modulus(T, nl), li = T < nl ? (uint64_t)T : 32 + (uint64_t)(T - nl);
                    const uint64_t sd = replay_seed + ((5ull << 32) | (uint64_t)((((pid * 32 + d) * 2 + k) * 64) + (int)li));
                    uint64_t *row = replay_rows.data() + (((size_t)d * 2 + k) * nt + T) * N;
                    for (int j = 0; j < N; j++) row[j] = splitmix_at(sd, (uint64_t)j) % q;
                }
                replay_rows_id = pid; replay_rows_level = level;
            }
            const uint64_t id = 1 + key_ids.size();
            HCR(hc_swk_load(hc, id, level, replay_rows.data()));
            key_ids[{ident, level}] = id; n_keys++;
            return id;
        }
        if (resnetReplaySeed()) {            // test mode: or_gen_swk(sk, gal, level, 7000003 seed + 131 gal + level) of the oracle network (tests/oracle_ckks.py Ckks.key): the
            const uint64_t seed = 7000003ull * resnetReplaySeed() + 131ull * gal + (uint64_t)level;      // oracle's errors drawn here, its splitmix rows and the arithmetic on the device
            std::vector<int64_t> es((size_t)beta * N), e;
            for (int dgt = 0; dgt < beta; dgt++) { replay::gauss(seed ^ (0xE44E44ull + (uint64_t)dgt * 7919), e); std::copy(e.begin(), e.end(), es.begin() + (long)dgt * N); }
            const uint64_t id = 1 + key_ids.size();
            HCR(hc_swk_generate_splitmix(hc, id, level, gal, d_sk, seed, es.data()));
            key_ids[{ident, level}] = id; n_keys++;
            return id;
        }
        // rlwe.GenSwitchingKey on the device (hc_swk_generate: ChaCha20 rows keyed by this bootstrapper's own stream, one Gaussian error per digit, the NTT of all
        // limbs in one batched launch): s_out = sigma_{gal^-1}(s) for a rotation / conjugation, s for relinearisation (gal = 0). Round 3 built every row on the
        // host and pushed it through one-row launches: 17-28 s per context, now a fraction of a second.
        if (!have_kg_seed) { for (int i = 0; i < 4; i++) { const uint64_t r = next(); kg_seed[2 * i] = (uint32_t)r; kg_seed[2 * i + 1] = (uint32_t)(r >> 32); } have_kg_seed = true; }
        const uint64_t id = 1 + key_ids.size();
        HCR(hc_swk_generate(hc, id, level, gal, d_sk, kg_seed));
        key_ids[{ident, level}] = id; n_keys++;
        return id;
    }
    uint32_t kg_seed[8]; bool have_kg_seed = false;
    uint64_t gal_rot(int k) const { const uint64_t twoN = 2ull * N; uint64_t e = (uint64_t)(int64_t)k & (twoN - 1), r = 1, b = 5; while (e) { if (e & 1) r = (r * b) % twoN; b = (b * b) % twoN; e >>= 1; } return r; }

    // ---------------- evaluator (ckks.Evaluator at any level)
    static void same_scale(double a, double b) { if (fabs(a / b - 1.0) > 1e-9) panic("scale mismatch in Add/Sub"); }
    DCt add(const DCt &a0, const DCt &b0) {
        const int L = std::min(a0.level, b0.level); same_scale(a0.scale, b0.scale);
        DCt r = new_ct(L, std::max(a0.deg, b0.deg), a0.scale);
        alg_ct += 3.0 * (r.deg + 1) * (L + 1);
        if (a0.deg == 1 && b0.deg == 1) { HCR(hc_lv_op2(hc, HC_LV_ADD, L, a0.p[0].get(), a0.p[1].get(), b0.p[0].get(), b0.p[1].get(), r.p[0].get(), r.p[1].get(), nullptr)); return r; }   // both polynomials per launch
        for (int d = 0; d <= r.deg; d++) {
            if (d <= a0.deg && d <= b0.deg) HCR(hc_lv_add(hc, L, a0.p[d].get(), b0.p[d].get(), r.p[d].get()));
            else copy_rows(r.p[d].get(), (d <= a0.deg ? a0 : b0).p[d].get(), (size_t)(L + 1));
        }
        return r;
    }
    DCt sub(const DCt &a0, const DCt &b0) {
        const int L = std::min(a0.level, b0.level); same_scale(a0.scale, b0.scale);
        if (a0.deg != b0.deg) panic("sub: degrees differ");
        DCt r = new_ct(L, a0.deg, a0.scale);
        alg_ct += 3.0 * (r.deg + 1) * (L + 1);
        if (r.deg == 1) HCR(hc_lv_op2(hc, HC_LV_SUB, L, a0.p[0].get(), a0.p[1].get(), b0.p[0].get(), b0.p[1].get(), r.p[0].get(), r.p[1].get(), nullptr));
        else for (int d = 0; d <= r.deg; d++) HCR(hc_lv_sub(hc, L, a0.p[d].get(), b0.p[d].get(), r.p[d].get()));
        return r;
    }
    std::vector<uint64_t> consts(const u128 mag, bool neg, int level) const {
        std::vector<uint64_t> c((size_t)level + 1);
        for (int l = 0; l <= level; l++) { uint64_t r = (uint64_t)(mag % Q[(size_t)l]); c[(size_t)l] = (neg && r) ? Q[(size_t)l] - r : r; }
        return c;
    }
    static void split_int(double x, u128 *mag, bool *neg) {      // int(round(x)) for |x| < 2^127
        *neg = x < 0; double a = fabs(x);
        double r = nearbyint(a);                                   // ties-to-even, as Python's round() on floats
        if (r < 18446744073709551616.0) *mag = (u128)(uint64_t)r;
        else { int e; double m = frexp(r, &e); *mag = ((u128)(uint64_t)ldexp(m, 64)) << (e - 64); }
    }
    DCt mul_const_int(const DCt &a, double k_rounded_value) {      // k given as an integral double
        u128 mag; bool neg; split_int(k_rounded_value, &mag, &neg);
        std::vector<uint64_t> c = consts(mag, neg, a.level);
        DCt r = new_ct(a.level, a.deg, a.scale); alg_ct += 2.0 * (a.deg + 1) * (a.level + 1);
        if (a.deg == 1) HCR(hc_lv_op2(hc, HC_LV_MUL_CONST, a.level, a.p[0].get(), a.p[1].get(), nullptr, nullptr, r.p[0].get(), r.p[1].get(), c.data()));
        else for (int d = 0; d <= a.deg; d++) HCR(hc_lv_mul_const(hc, a.level, a.p[d].get(), c.data(), r.p[d].get()));
        return r;
    }
    // sum_t k_t * a_t (+ c0) at `level` as ONE launch (hc_lv_lincomb2): what a leaf of the polynomial evaluators computes with a MultByConst per power and an Add chain.
    // ks: integral doubles (as mul_const_int takes them); scale: the label of the result
    DCt lincomb(const std::vector<DCt> &as, const std::vector<double> &ks, int level, double scale, bool with_c0 = false, double c0 = 0) {
        const size_t nt_ = as.size();
        if (nt_ < 1 || nt_ > 8 || ks.size() != nt_) panic("lincomb: 1..8 terms");
        std::vector<uint64_t> cs(nt_ * (size_t)(level + 1)), cadd; std::vector<const uint64_t *> p0(nt_), p1(nt_);
        for (size_t t = 0; t < nt_; t++) {
            if (as[t].deg != 1 || as[t].level < level) panic("lincomb: degree-1 operands at or above the level");
            u128 mag; bool neg; split_int(ks[t], &mag, &neg);
            std::vector<uint64_t> c = consts(mag, neg, level); std::copy(c.begin(), c.end(), cs.begin() + (long)(t * (size_t)(level + 1)));
            p0[t] = as[t].p[0].get(); p1[t] = as[t].p[1].get();
        }
        if (with_c0) { u128 mag; bool neg; split_int(c0, &mag, &neg); cadd = consts(mag, neg, level); }
        DCt r = new_ct(level, 1, scale); alg_ct += (2.0 * (double)nt_ + 2.0) * (level + 1);
        HCR(hc_lv_lincomb2(hc, level, (int)nt_, p0.data(), p1.data(), cs.data(), with_c0 ? cadd.data() : nullptr, r.p[0].get(), r.p[1].get()));
        return r;
    }
    DCt add_const_int(const DCt &a, double k_rounded_value) {
        u128 mag; bool neg; split_int(k_rounded_value, &mag, &neg);
        std::vector<uint64_t> c = consts(mag, neg, a.level);
        DCt r = a; r.p[0] = block(); alg_ct += 2.0 * (a.level + 1);
        HCR(hc_lv_add_const(hc, a.level, a.p[0].get(), c.data(), r.p[0].get()));
        return r;
    }
    // evaluator.AddConst with a real constant: scaleUpExact(c, scale, q) = floor(|c * scale| + 0.5), sign restored modulo q (the AddConst digests of
    // tests/golden/ref_trace_cheby_5_1.json pin the rule)
    DCt add_const(const DCt &a, double c) { const double k = floor(fabs(a.scale * c) + 0.5); return add_const_int(a, c < 0 ? -k : k); }
    // evaluator.MultByConst with a float64: a constant with a fractional part is carried times q_level (set_scale below uses the same rule)
    DCt mul_const_float(const DCt &a, double c) {
        const double mult = c - (double)(int64_t)c != 0 ? (double)Q[(size_t)a.level] : 1.0;
        DCt r = mul_const_int(a, floor(fabs(c * mult) + 0.5) * (c < 0 ? -1.0 : 1.0)); r.scale = a.scale * mult;
        return r;
    }
    DCt mul_plain(const DCt &a, const DPt &pt) {
        if (pt.level < a.level) panic("mul_plain: plaintext below the ciphertext's level");
        DCt r = new_ct(a.level, a.deg, a.scale * pt.scale); alg_ct += 2.0 * (a.deg + 1) * (a.level + 1); alg_shared += a.level + 1;
        if (a.deg == 1) HCR(hc_lv_op2(hc, HC_LV_MUL_PLAIN, a.level, a.p[0].get(), a.p[1].get(), pt.p.get(), nullptr, r.p[0].get(), r.p[1].get(), nullptr));
        else for (int d = 0; d <= a.deg; d++) HCR(hc_lv_mul_plain(hc, a.level, a.p[d].get(), pt.p.get(), r.p[d].get()));
        return r;
    }
    DCt mul_by_i(const DCt &a) {
        DCt r = new_ct(a.level, a.deg, a.scale); alg_ct += 2.0 * (a.deg + 1) * (a.level + 1); alg_shared += a.level + 1;
        if (a.deg == 1) HCR(hc_lv_op2(hc, HC_LV_MUL_PLAIN, a.level, a.p[0].get(), a.p[1].get(), mono_i.get(), nullptr, r.p[0].get(), r.p[1].get(), nullptr));
        else for (int d = 0; d <= a.deg; d++) HCR(hc_lv_mul_plain(hc, a.level, a.p[d].get(), mono_i.get(), r.p[d].get()));
        return r;
    }
    DCt mul_relin(const DCt &a, const DCt &b) {                   // evaluator.MulRelin: tensor, key switch of c2 with the rlk
        const int L = std::min(a.level, b.level);
        DCt r = new_ct(L, 1, a.scale * b.scale); alg_ct += 6.0 * (L + 1); alg_shared += ks_rows(L);
        auto d1 = block(), d2 = block();
        HCR(hc_lv_mul_tensor(hc, L, a.p[0].get(), a.p[1].get(), b.p[0].get(), b.p[1].get(), r.p[0].get(), d1.get(), d2.get()));
        HCR(hc_keyswitch_add(hc, key(0, L), L, d2.get(), r.p[0].get(), d1.get(), r.p[0].get(), r.p[1].get())); n_keyswitch++;      // (d0 + ks0, d1 + ks1) inside ModDown's last pass
        return r;
    }
    // MulRelin followed by one Rescale: the key switch's ModDown and the rescale share one forward transform per limb (hc_keyswitch_add_rescale); the residues of the two steps
    DCt mul_relin_rescale(const DCt &a, const DCt &b) {
        const int L = std::min(a.level, b.level);
        static const bool split = getenv("HCONV_NO_FUSED_RESCALE") != nullptr;
        if (L < 2 || split) return rescale(mul_relin(a, b));
        DCt r = new_ct(L - 1, 1, a.scale * b.scale / (double)Q[(size_t)L]); alg_ct += 6.0 * (L + 1) + 2.0 * (2 * L + 1); alg_shared += ks_rows(L);
        auto d0 = block(), d1 = block(), d2 = block();
        HCR(hc_lv_mul_tensor(hc, L, a.p[0].get(), a.p[1].get(), b.p[0].get(), b.p[1].get(), d0.get(), d1.get(), d2.get()));
        HCR(hc_keyswitch_add_rescale(hc, key(0, L), L, d2.get(), d0.get(), d1.get(), r.p[0].get(), r.p[1].get())); n_keyswitch++;
        return r;
    }
    DCt rescale(const DCt &a) {                                     // one DivRoundByLastModulusNTT
        if (a.level < 1) panic("rescale at level 0");
        DCt r = new_ct(a.level - 1, a.deg, a.scale / (double)Q[(size_t)a.level]); alg_ct += (double)(a.deg + 1) * (2 * a.level + 1);
        if (a.deg == 1) HCR(hc_div_round_last2(hc, a.level, a.p[0].get(), a.p[1].get(), r.p[0].get(), r.p[1].get()));     // both polynomials per launch
        else for (int d = 0; d <= a.deg; d++) HCR(hc_div_round_last(hc, a.level, a.p[d].get(), r.p[d].get()));
        return r;
    }
    DCt galois(const DCt &a, uint64_t gal) {                       // evaluator.permuteNTT: key switch c1, + c0, permute both
        const int L = a.level;
        DCt r = new_ct(L, 1, a.scale); alg_ct += 4.0 * (L + 1); alg_shared += ks_rows(L);
        HCR(hc_keyswitch_rotate(hc, key(gal, L), gal, L, a.p[0].get(), a.p[1].get(), r.p[0].get(), r.p[1].get(), 0)); n_keyswitch++;   // + c0 and the permutation inside ModDown's last pass
        return r;
    }
    DCt rotate(const DCt &a, int k) { k = ((k % n) + n) % n; return k == 0 ? a : galois(a, gal_rot(k)); }
    DCt conjugate(const DCt &a) { return galois(a, 2ull * N - 1); }
    DCt mod_raise(const DCt &a, int level) {                        // ckks.(*Bootstrapper).modUp
        if (a.level != 0) panic("mod_raise expects a level-0 ciphertext");
        DCt r = new_ct(level, 1, a.scale); alg_ct += 2.0 + 2.0 * (level + 1);
        for (int d = 0; d < 2; d++) HCR(hc_lv_mod_raise(hc, level, a.p[d].get(), r.p[d].get()));
        return r;
    }
    // evaluator.SetScale: MultByConst(scale / ct.Scale) — a constant with a fractional part is carried times
    // q_level — then Rescale(scale) (drop while scale/q_L >= scale/2), then the scale is forced
    DCt set_scale(const DCt &a, double scale) {
        const double c = scale / a.scale; DCt r = a;
        if (c != 1.0) {
            double mult = 1.0; if (c - (double)(int64_t)c != 0) mult = (double)Q[(size_t)a.level];
            r = mul_const_int(a, floor(fabs(c * mult) + 0.5) * (c < 0 ? -1.0 : 1.0)); r.scale = a.scale * mult;
        }
        while (r.level > 0 && r.scale / (double)Q[(size_t)r.level] >= scale / 2) r = rescale(r);
        r.scale = scale;
        return r;
    }
    // debugging aid (HCONV_DEBUG_BOOT): decrypt on the limbs 0, 1 (needs |message| * scale < Q0 Q1 / 2) and decode to slots
    std::vector<cplx> debug_slots(const DCt &a) {
        Single one(this);                                          // image 0 of a batch
        const int L = std::min(a.level, 1); auto t = block();
        HCR(hc_lv_mul_plain(hc, L, a.p[1].get(), d_sk, t.get())); HCR(hc_lv_add(hc, L, a.p[0].get(), t.get(), t.get())); HCR(hc_lv_intt(hc, L, t.get(), t.get()));
        std::vector<uint64_t> m((size_t)(L + 1) * N); HCR(hc_download(hc, m.data(), t.get(), m.size() * 8));
        std::vector<double> cf((size_t)N);
        if (L == 0) { const uint64_t q0 = Q[0]; for (int j = 0; j < N; j++) cf[(size_t)j] = (m[(size_t)j] > q0 / 2 ? -(double)(q0 - m[(size_t)j]) : (double)m[(size_t)j]) / a.scale; }
        else {
            const uint64_t q0 = Q[0], q1 = Q[1]; uint64_t inv = 1; { uint64_t b = q0 % q1, e = q1 - 2; while (e) { if (e & 1) inv = mulmod(inv, b, q1); b = mulmod(b, b, q1); e >>= 1; } }
            const u128 QQ = (u128)q0 * q1;
            for (int j = 0; j < N; j++) { const uint64_t a0 = m[(size_t)j], a1 = m[(size_t)N + j], d = (a1 % q1 + q1 - a0 % q1) % q1; const u128 x = (u128)a0 + (u128)q0 * mulmod(d, inv, q1);
                cf[(size_t)j] = x > QQ / 2 ? -(double)(QQ - x) / a.scale : (double)x / a.scale; }
        }
        std::vector<cplx> v((size_t)N / 2); for (int i = 0; i < N / 2; i++) v[(size_t)i] = cplx(cf[(size_t)i], cf[(size_t)(i + N / 2)]);
        enc.fft(v);
        return v;
    }
    static void debug_compare(const char *what, const std::vector<cplx> &want, const std::vector<cplx> &got) {
        double mx = 0, sum = 0, mag = 0; for (size_t i = 0; i < want.size(); i++) { const double e = std::abs(want[i] - got[i]); mx = std::max(mx, e); sum += e; mag = std::max(mag, std::abs(want[i])); }
        fprintf(stderr, "[debug] %s: max |want| %.4g, error max 2^%.2f mean 2^%.2f\n", what, mag, log2(mx), log2(sum / (double)want.size()));
    }
    static DCt relabel(const DCt &a, double scale) { if (fabs(a.scale / scale - 1.0) > 1e-6) panic("relabel: scales are not close"); DCt r = a; r.scale = scale; return r; }

    // ---------------- encoding
    // a 0/1 slot mask at (level, scale): encoded once per context (the reference re-encodes them in every layer, conv.go:423, 381)
    DPt encode_mask(const std::string &key, const std::vector<int> &idx, int level, double scale) {
        const std::string k = key + "/" + std::to_string(level);
        auto it = pt_cache.find(k); if (it != pt_cache.end()) return it->second;
        std::vector<cplx> tmp((size_t)N / 2, cplx(0, 0)); for (size_t i = 0; i < idx.size(); i++) tmp[i] = cplx((double)idx[i], 0);
        return pt_cache.emplace(k, encode(tmp, level, scale)).first->second;
    }
    DPt encode(const std::vector<cplx> &slots, int level, double scale) {
        std::vector<uint64_t> rows = enc.Encode(slots, scale, Q.data(), level + 1);
        Single one(this);
        DPt pt; pt.level = level; pt.scale = scale; pt.p = block1();
        pack_rows(rows, level + 1);
        HCR(hc_upload(hc, pt.p.get(), rows.data(), rows.size() * 8));
        HCR(hc_lv_ntt(hc, level, pt.p.get(), pt.p.get()));
        return pt;
    }

    // a diagonal as the reference's encodeDiagonal leaves it (minus the Montgomery factor): mod Q_0..Q_level and mod every P, NTT domain
    DPt encode_qp(const std::vector<cplx> &slots, int level, double scale) {
        const int nl = level + 1, np = (int)P.size();
        std::vector<uint64_t> mods(Q.begin(), Q.begin() + nl); mods.insert(mods.end(), P.begin(), P.end());
        std::vector<uint64_t> rows;
        if ((int)slots.size() == N / 2) rows = enc.Encode(slots, scale, mods.data(), nl + np);
        else {                                  // a sparse-slot diagonal: the encoder's sparse embedding (pinned: ref_trace_diag_sparse_ls*.json)
            int lg = 0; while ((1 << lg) < (int)slots.size()) lg++;
            auto it = sub_enc.find(lg + 1); if (it == sub_enc.end()) it = sub_enc.emplace(lg + 1, Encoder(lg + 1)).first;
            rows = enc.EncodeSparse(it->second, slots, scale, mods.data(), nl + np);
        }
        Single one(this);
        DPt pt; pt.level = level; pt.scale = scale;
        { void *v = nullptr; HCR(hc_malloc(hc, rows.size() * 8, &v)); uint64_t *d = (uint64_t *)v; hc_ctx *h = hc; pt.p = std::shared_ptr<uint64_t>(d, [h](uint64_t *x) { hc_free(h, x); }); }
        pack_rows(rows, nl);
        HCR(hc_upload(hc, pt.p.get(), rows.data(), rows.size() * 8));
        HCR(hc_lv_ntt(hc, level, pt.p.get(), pt.p.get()));
        for (int j = 0; j < np; j++) { uint64_t *r = pt.p.get() + (size_t)(nl + j) * N; HCR(hc_ntt(hc, NQ + j, r, r, 1)); }
        return pt;
    }

    // ---------------- DFT matrices in diagonal form (the encoder's own butterflies, no bit reversal)
    // `E`: encoder of the ring the DFT belongs to (the full ring, or the subring X^(2^ls) of sparse packing: same butterflies
    // with that ring's roots, tiled over the n full slots); rotation indices modulo `period` (the slot vector's period)
    DiagMat dft_stage(int ln, bool inverse, const Encoder &E, int period) const {
        const int lenh = ln >> 1, lenq = ln << 2, gap = 2 * E.Nn / lenq;
        std::vector<cplx> d0((size_t)n), dp((size_t)n, cplx(0, 0)), dm((size_t)n, cplx(0, 0));
        for (int p = 0; p < n; p++) {
            const int j = p % ln; const bool first = j < lenh; const int jj = first ? j : j - lenh;
            const int idx = inverse ? (lenq - (E.rotGroup[(size_t)jj] % lenq)) * gap : (E.rotGroup[(size_t)jj] % lenq) * gap;
            const cplx w = E.roots[(size_t)idx];
            if (inverse) { d0[(size_t)p] = first ? cplx(1, 0) : -w; if (first) dp[(size_t)p] = cplx(1, 0); else dm[(size_t)p] = w; }
            else { d0[(size_t)p] = first ? cplx(1, 0) : -w; if (first) dp[(size_t)p] = w; else dm[(size_t)p] = cplx(1, 0); }
        }
        DiagMat M; M[0] = d0;
        const int kp = lenh % period, km = ((-lenh) % period + period) % period;
        auto acc = [&](int k, const std::vector<cplx> &d) { auto it = M.find(k); if (it == M.end()) M[k] = d; else for (int p = 0; p < n; p++) it->second[(size_t)p] += d[(size_t)p]; };
        acc(kp, dp); acc(km, dm);
        return M;
    }
    DiagMat matmul_diag(const DiagMat &M2, const DiagMat &M1, int period) const {        // M2 . M1 (M1 applied first)
        DiagMat out;
        for (auto &e2 : M2) for (auto &e1 : M1) {
            const int k2 = e2.first, k = (e1.first + k2) % period;
            auto it = out.find(k); if (it == out.end()) it = out.emplace(k, std::vector<cplx>((size_t)n, cplx(0, 0))).first;
            for (int p = 0; p < n; p++) it->second[(size_t)p] += e2.second[(size_t)p] * e1.second[(size_t)((p + k2) % n)];
        }
        for (auto it = out.begin(); it != out.end();) { bool nz = false; for (auto &v : it->second) if (v != cplx(0, 0)) { nz = true; break; } if (nz) ++it; else it = out.erase(it); }
        return out;
    }
    static std::vector<int> fit(std::vector<int> g, int logn) {
        while (std::accumulate(g.begin(), g.end(), 0) > logn) (*std::max_element(g.begin(), g.end()))--;
        return g;
    }
    std::vector<DiagMat> dft_groups(bool inverse, std::vector<int> sizes, double constant, int ls) const {
        const int logn = LOGN - 1 - ls, ns = n >> ls;
        sizes = fit(sizes, logn);
        Encoder sub(LOGN - ls); const Encoder &E = ls ? sub : enc;
        std::vector<int> lens; for (int s = 0; s < logn; s++) lens.push_back(inverse ? ns >> s : 2 << s);
        const double c = pow(constant, 1.0 / (double)sizes.size());
        std::vector<DiagMat> groups; size_t pos = 0;
        for (int gs : sizes) {
            DiagMat M; bool have = false;
            for (int i = 0; i < gs; i++, pos++) { DiagMat S = dft_stage(lens[pos], inverse, E, ns); M = have ? matmul_diag(S, M, ns) : S; have = true; }
            for (auto &e : M) for (auto &v : e.second) v *= c;
            groups.push_back(std::move(M));
        }
        return groups;
    }
    // ---------------- the same matrices as the reference's Lattigo fork builds them (full slots, ls = 0)
    // ckks.(*Bootstrapper).genDFTMatrices -> GenCoeffsToSlotsMatrix / GenSlotsToCoeffsMatrix -> computeDFTMatrices (fftPlainVec /
    // fftInvPlainVec, genFFTDiagMatrix, multiplyFFTMatrixWithNextFFTLevel) of github.com/dwkim606/test_lattigo (binary only): the
    // radix-2 levels as (a, b, c) diagonals, merged ceil(remaining / depth-left) at a time, every complex product evaluated as Go does
    // (four rounded products, no fused multiply-add; build flags carry no -march, so the compiler has no FMA to contract into) and the
    // sums in the fork's order; roots = the encoder's table (computeRoots(2n) is the same Go cos / sin of the same angles).
    // tests/golden/ref_trace_diag_5_1.json holds the SHA-256 of every vector the binary hands to its encoder and of every encoded
    // polynomial; HCONV_DFT_DIGESTS=<file> makes this host write its own (tests/test_emu.py, tests/test_gpu_z_cli.py compare).
    static cplx go_mul(cplx x, cplx y) { return cplx(x.real() * y.real() - x.imag() * y.imag(), x.real() * y.imag() + x.imag() * y.real()); }
    static void add_to(DiagMat &M, int k, std::vector<cplx> v) {
        auto it = M.find(k);
        if (it == M.end()) M.emplace(k, std::move(v)); else for (size_t p = 0; p < v.size(); p++) it->second[p] += v[p];
    }
    // ls > 0 (sparse slots, round 3): logSlots = LOGN-1-ls, vectors of dslots = 2 * 2^logSlots entries (fftPlainVec fills both halves),
    // SlotsToCoeffs' first matrix = genWfftRepack (the (re | im) -> re + i im map: diagonals 0 and 2^logSlots) merged with its DFT levels
    // (rotations modulo dslots), CoeffsToSlots' last matrix zeroed on its upper half: computeDFTMatrices' repacking branches, pinned against
    // the binary by tests/golden/ref_trace_diag_sparse_ls*.json (gotrace -diag -logslots K).
    std::vector<DiagMat> lattigo_dft(bool inverse, int depth, double diffscale, int ls = 0) const {
        const int logn = LOGN - 1 - ls, ns = n >> ls, ds = ls ? 2 * ns : ns, size = ls ? 2 : 1;
        std::vector<int> pow5((size_t)(2 * ns + 1), 1);
        for (size_t i = 1; i < pow5.size(); i++) pow5[i] = (int)(((long)pow5[i - 1] * 5) & (4L * ns - 1));
        const int root_gap = n / ns;                             // computeRoots(2 ns)[k] = the full ring's root of the same angle
        std::vector<std::vector<cplx>> a, b, c;
        for (int s = 0; s < logn; s++) {
            const int m = inverse ? ns >> s : 2 << s, tt = m >> 1, gap = ns / m, mask = (m << 2) - 1;
            std::vector<cplx> va((size_t)ds, cplx(0, 0)), vb((size_t)ds, cplx(0, 0)), vc((size_t)ds, cplx(0, 0));
            for (int i = 0; i < ns; i += m) for (int j = 0; j < tt; j++) {
                const int k = inverse ? ((m << 2) - (pow5[(size_t)j] & mask)) * gap : (pow5[(size_t)j] & mask) * gap;
                const cplx w = enc.roots[(size_t)k * (size_t)root_gap];
                for (int u = 0; u < size; u++) {
                    const size_t i1 = (size_t)(i + j + u * ns), i2 = (size_t)(i + j + tt + u * ns);
                    va[i1] = cplx(1, 0); va[i2] = -w;
                    if (inverse) { vb[i1] = cplx(1, 0); vc[i2] = w; }
                    else { vb[i1] = w; vc[i2] = cplx(1, 0); }
                }
            }
            a.push_back(std::move(va)); b.push_back(std::move(vb)); c.push_back(std::move(vc));
        }
        std::vector<int> merge((size_t)depth, 0);
        for (int i = 0, lvl = logn; i < depth; i++) { const int d = (lvl + depth - i - 1) / (depth - i); merge[(size_t)(inverse ? i : depth - i - 1)] = d; lvl -= d; }
        auto rot_of = [&](int level) { return inverse ? 1 << (level - 1) : 1 << (logn - level); };
        auto rotated_times = [&](const std::vector<cplx> &v, int r, const std::vector<cplx> &w) {       // mul(rotate(v, r), w): rotate is to the left, over the vector's own length
            std::vector<cplx> out((size_t)ds); for (int p = 0; p < ds; p++) out[(size_t)p] = go_mul(v[(size_t)((p + r) & (ds - 1))], w[(size_t)p]); return out; };
        auto times_next = [&](const DiagMat &M, int nmod, int nxt) {
            const int r = rot_of(nxt) & (nmod - 1), x = logn - nxt;
            DiagMat nw;
            for (auto &e : M) {           // the fork ranges over a Go map here; at every position at most two of the three terms are non-zero, so the sums do not depend on the order
                add_to(nw, e.first, rotated_times(e.second, 0, a[(size_t)x]));
                add_to(nw, (e.first + r) & (nmod - 1), rotated_times(e.second, r, b[(size_t)x]));
                add_to(nw, (e.first - r) & (nmod - 1), rotated_times(e.second, ds - r, c[(size_t)x]));
            }
            return nw;
        };
        std::vector<DiagMat> out;
        for (int i = 0, lvl = logn; i < depth; i++) {
            DiagMat M; int nmod = ns;
            if (ls && !inverse && i == 0) {      // genWfftRepack, merged with the first DFT level
                DiagMat W; W[0].assign((size_t)ds, cplx(0, 0)); W[ns].assign((size_t)ds, cplx(0, 0));
                for (int p = 0; p < ns; p++) { W[0][(size_t)p] = cplx(1, 0); W[0][(size_t)(p + ns)] = cplx(0, 1); W[ns][(size_t)p] = cplx(0, 1); W[ns][(size_t)(p + ns)] = cplx(1, 0); }
                nmod = ds;
                M = times_next(W, nmod, lvl);
            } else { const int r = rot_of(lvl), x = logn - lvl; add_to(M, 0, a[(size_t)x]); add_to(M, r, b[(size_t)x]); add_to(M, ns - r, c[(size_t)x]); }
            for (int j = 0, nxt = lvl - 1; j < merge[(size_t)i] - 1; j++, nxt--) M = times_next(M, nmod, nxt);
            out.push_back(std::move(M)); lvl -= merge[(size_t)i];
        }
        if (ls && inverse) for (auto &e : out.back()) for (int p = ns; p < ds; p++) e.second[(size_t)p] = cplx(0, 0);       // repacking after CoeffsToSlots
        for (auto &M : out) for (auto &e : M) for (auto &v : e.second) v = go_mul(v, cplx(diffscale, 0));
        return out;
    }
    // findbestbabygiantstepsplit / bsgsIndex of the fork (maxN1N2Ratio = 16): the first N1 with more hoisted (baby) rotations than giant
    // ones, doubled until their ratio reaches 16
    static int lattigo_n1(const DiagMat &M, int slots, double max_ratio = 16.0) {
        for (int n1 = 1; n1 < slots; n1 <<= 1) {
            std::map<int, int> index; for (auto &e : M) index[(e.first & (slots - 1)) / n1]++;
            if (!index.count(0)) continue;
            int hoisted = index[0] - 1, normal = (int)index.size() - 1;
            if (normal == 0) return n1 / 2;
            if (hoisted > normal) {
                while ((double)hoisted / (double)normal < max_ratio) { if (normal / 2 == 0) break; n1 *= 2; hoisted = hoisted * 2 + 1; normal /= 2; }
                return n1;
            }
        }
        return 1;
    }
    FILE *dft_digests = nullptr;
    // slots: the length of M's vectors (n, or 2^(logSlots+1) for a sparse-slot matrix: rotations modulo it, sparse embedding)
    LT plan(const DiagMat &M, int level, double pt_scale, bool lattigo_split = false, const char *tag = "", int slots = 0) {          // BSGS split + the pre-rotated, encoded diagonals
        LT lt; lt.level = level; lt.pt_scale = pt_scale;
        const int n = slots ? slots : this->n;
        int best = -1;
        for (int n1 = 1; n1 <= n; n1 <<= 1) {
            std::set<int> babies, giants; for (auto &e : M) { babies.insert(e.first % n1); giants.insert(e.first - e.first % n1); }
            babies.erase(0); giants.erase(0);
            const int cost = (int)(babies.size() + giants.size());
            if (best < 0 || cost < best) { best = cost; lt.n1 = n1; }
        }
        if (lattigo_split) lt.n1 = lattigo_n1(M, n);
        for (auto &e : M) {
            const int k = e.first, g = k - k % lt.n1, b = k % lt.n1;
            std::vector<cplx> rolled((size_t)n); for (int p = 0; p < n; p++) rolled[(size_t)p] = e.second[(size_t)(((p - g) % n + n) % n)];     // np.roll(diag, g) = the fork's rotate(v, -N1*j)
            lt.giant[g][b] = encode_qp(rolled, level, pt_scale);      // every bootstrapper's linear transforms run in the extended basis (linear_transform_qp)
            lt.qp = true;
            if (dft_digests) {      // what the reference's encodeDiagonal receives and returns (mod Q): values; NTT rows in Montgomery form + the spare zero limb
                Sha256 hv; hv.update(rolled.data(), rolled.size() * sizeof(cplx));
                std::vector<uint64_t> rows((size_t)(level + 1) * N), zero((size_t)N, 0);
                HCR(hc_download(hc, rows.data(), lt.giant[g][b].p.get(), rows.size() * 8));
                unpack_rows(rows, level + 1);
                for (int l = 0; l <= level; l++) { const uint64_t q = Q[(size_t)l], r = (uint64_t)((((u128)1) << 64) % q); for (int j = 0; j < N; j++) rows[(size_t)l * N + j] = mulmod(rows[(size_t)l * N + j], r, q); }
                Sha256 hq; hq.update(rows.data(), rows.size() * 8); hq.update(zero.data(), zero.size() * 8);
                std::string mp = "";
                if (lt.qp) {            // ... and mod P
                    const int np = (int)P.size(); std::vector<uint64_t> prow((size_t)np * N);
                    HCR(hc_download(hc, prow.data(), lt.giant[g][b].p.get() + (size_t)(level + 1) * N, prow.size() * 8));
                    for (int j = 0; j < np; j++) { const uint64_t q = P[(size_t)j], r = (uint64_t)((((u128)1) << 64) % q); for (int i = 0; i < N; i++) prow[(size_t)j * N + i] = mulmod(prow[(size_t)j * N + i], r, q); }
                    Sha256 hp; hp.update(prow.data(), prow.size() * 8); mp = hp.hex();
                }
                fprintf(dft_digests, "{\"matrix\": \"%s\", \"chain\": %d, \"level\": %d, \"scale\": %.17g, \"N1\": %d, \"k\": %d, \"values\": \"%s\", \"mQ\": \"%s\", \"mP\": \"%s\"}\n", tag, chain, level, pt_scale, lt.n1, k, hv.hex().c_str(), hq.hex().c_str(), mp.c_str());
            }
        }
        return lt;
    }
    // ckks.(*evaluator).LinearTransform -> MultiplyByDiagMatrixBSGS exactly as the reference's fork computes it (test_run @52a580; tests/lattigo_lt.py
    // is the same algorithm on the oracle and reproduces the binary's ModDown inputs / outputs and result on planted data, tests/test_oracle_pin_lt.py;
    // tests/oracle_ckks.py Ckks.linear_transform_qp is this function on the oracle chain). Everything stays in the extended basis QP until a
    // complete sum exists: the baby-step rotations are key-switched WITHOUT the division by P on one digit decomposition and get P*c0 added
    // (rot_i = (phi_i(P c0 + d0_i), phi_i(d1_i)) mod QP); per giant step j != 0 their products with the diagonals (encoded mod Q and mod P) are
    // summed in QP and brought down ONCE, a rotation-0 diagonal multiplies the input itself after that division, the second component is
    // key-switched again without ModDown and permuted into QP accumulators, the first is permuted straight into the result; giant step 0 adds
    // its products to the same accumulators, which are brought down once. 2 ModDowns per giant step + 2 instead of 2 per baby step.
    // hoist_c1 / hoist_level: the reference's behaviour on a ciphertext ABOVE the matrix level (the stock Bootstrapp's last SlotsToCoeffs matrix: ciphertext level 15,
    // matrix level 14). DecomposeNTT runs at the matrix level, but rotateHoistedNoModDown takes its level from the ciphertext (test_run @0x524d60), so the hoisted key
    // switch runs one digit further than was decomposed and reads what the evaluator's decomposition pool still holds there: the last digit of the PREVIOUS
    // LinearTransform's input. Harmless for the value (that digit's gadget factor vanishes modulo the limbs kept), but the residues depend on it; the same residues come
    // out of ONE decomposition at the ciphertext's level of hoist_c1 = (c1's limbs up to the matrix level | the previous input's limbs above it), rows above the matrix
    // level dropped afterwards (tests/oracle_ckks.py linear_transform_qp, pinned by ref_trace_chain_bl_5_1.json).
    // fuse_min_scale > 0: the caller rescales next by ckks Rescale's drop rule with this minimum scale; when its first drop is due, it rides in the final ModDown
    // (hc_mod_down2_add_rescale) and the result comes back one level down
    DCt linear_transform_qp(const DCt &ct, const LT &lt, const uint64_t *hoist_c1 = nullptr, int hoist_level = -1, double fuse_min_scale = 0) {
        const int L = ct.level, nl = L + 1, np = (int)P.size(), nt = nl + np; const size_t zs = (size_t)nt * N;
        std::map<int, std::vector<int>> index; std::set<int> babies;
        for (auto &g : lt.giant) for (auto &b : g.second) { index[g.first / lt.n1].push_back(b.first); if (b.first) babies.insert(b.first); }
        const int Lb = hoist_c1 ? hoist_level : L;                                  // the level the baby-step key switches run at
        alg_ct += 4.0 * nl;
        for (auto &g : lt.giant) { if (g.first) alg_shared += ks_rows(L); alg_shared += (double)g.second.size() * nt; }
        alg_shared += (double)babies.size() * ks_rows(Lb);
        for (int b : babies) key(gal_rot(b), Lb, 1);                                // key generation (if any) before a decomposition is taken
        for (auto &g : lt.giant) if (g.first) key(gal_rot(g.first), L, 2);
        std::vector<uint64_t> pmod((size_t)nl), zeros((size_t)nl, 0);
        for (int l = 0; l < nl; l++) { uint64_t r = 1; for (uint64_t pj : P) r = mulmod(r, pj % Q[(size_t)l], Q[(size_t)l]); pmod[(size_t)l] = r; }
        auto pc0 = block(); HCR(hc_lv_mul_const(hc, L, ct.p[0].get(), pmod.data(), pc0.get()));                       // P * c0
        std::map<int, std::shared_ptr<uint64_t>> rot;
        if (!babies.empty()) {
            const uint64_t *cx = hoist_c1 ? hoist_c1 : ct.p[1].get();
            HCR(hc_keyswitch_decompose(hc, Lb, cx));
            auto wide = hoist_c1 ? block_qp2() : std::shared_ptr<uint64_t>();
            if (hoist_c1 && nb != 1) panic("linear_transform_qp: the stale-digit hoisting of the stock Bootstrapp runs one image at a time");
            if (!hoist_c1) {                                                                                            // all baby steps: inner products sharing the digit reads, + P c0, permutation
                std::vector<uint64_t> ids, gals; std::vector<uint64_t *> outs;
                for (int b : babies) { const uint64_t gal = gal_rot(b); auto r = block_qp2(); ids.push_back(key(gal, L, 1)); gals.push_back(gal); outs.push_back(r.get()); rot[b] = r; n_keyswitch++; }
                static const bool one_by_one = getenv("HCONV_ROTATE_ONE_BY_ONE") != nullptr;
                if (!one_by_one) HCR(hc_keyswitch_qp_rotate_many(hc, (int)ids.size(), ids.data(), gals.data(), L, pc0.get(), cx, outs.data()));
                else for (size_t i = 0; i < ids.size(); i++) HCR(hc_keyswitch_qp_rotate(hc, ids[i], gals[i], L, pc0.get(), cx, outs[i], 1, 0));
            } else for (int b : babies) {
                const uint64_t gal = gal_rot(b);
                auto r = block_qp2();
                {                                                                                                  // [2][Lb+1+np][N] -> rows 0..L and the P rows of each component
                    auto acc = block_qp2();
                    HCR(hc_keyswitch_qp(hc, key(gal, Lb, 1), Lb, cx, wide.get(), 1));
                    const size_t zw = (size_t)(Lb + 1 + np) * N;
                    for (int k = 0; k < 2; k++) {
                        HCR(hc_copy(hc, acc.get() + (size_t)k * zs, wide.get() + (size_t)k * zw, (size_t)nl * N * 8));
                        HCR(hc_copy(hc, acc.get() + (size_t)k * zs + (size_t)nl * N, wide.get() + (size_t)k * zw + (size_t)(Lb + 1) * N, (size_t)np * N * 8));
                    }
                    HCR(hc_lv_add(hc, L, acc.get(), pc0.get(), acc.get()));                                              // the Q rows of the first component
                    HCR(hc_qp_permute2(hc, gal, L, acc.get(), r.get()));
                }
                n_keyswitch++;
                rot[b] = r;
            }
        }
        DCt res = new_ct(L, 1, ct.scale * lt.pt_scale); bool have_res[2] = {false, false};
        auto add_to_res = [&](int k, const std::shared_ptr<uint64_t> &x) { if (have_res[k]) HCR(hc_lv_add(hc, L, res.p[k].get(), x.get(), res.p[k].get())); else { res.p[k] = x; have_res[k] = true; } };
        auto B = block_qp2(); bool haveB = false;
        // the diagonal sums of all giant steps first, THREE per pass over the rotations (hc_qp_mul_sum_many, up to four: the rotated ciphertexts - the largest operands of a linear transform - are
        // read once per group instead of once per giant step); giant step 0's sum lands in the accumulators B, which the other giant steps' key switches then add to
        // (modular sums commute: the residues of the reference's order)
        struct Sum { uint64_t *out; std::map<int, const uint64_t *> pt; };
        std::vector<Sum> sums; std::map<int, std::shared_ptr<uint64_t>> Aof;
        if (index.count(0)) { Sum s0; s0.out = B.get(); for (int i : index[0]) if (i) s0.pt[i] = lt.giant.at(0).at(i).p.get(); if (!s0.pt.empty()) { sums.push_back(s0); haveB = true; } }
        for (auto &ix : index) {
            const int j = ix.first; if (j == 0) continue;
            const auto &row = lt.giant.at(j * lt.n1);
            Sum sj; for (int i : ix.second) if (i) sj.pt[i] = row.at(i).p.get();
            if (!sj.pt.empty()) { Aof[j] = block_qp2(); sj.out = Aof[j].get(); sums.push_back(sj); }
        }
        static const int sums_per_pass = getenv("HCONV_SUMS_PER_PASS") ? std::max(1, std::min(4, atoi(getenv("HCONV_SUMS_PER_PASS")))) : 3;      // A/B switch; 1 = one pass per giant step. Measured (profiles/round4_chain_occupancy_ab.txt, round 12): 3 per pass
        for (size_t u = 0; u < sums.size(); u += (size_t)sums_per_pass) {
            const size_t ng = std::min((size_t)sums_per_pass, sums.size() - u);
            std::set<int> un; for (size_t v = u; v < u + ng; v++) for (auto &kv : sums[v].pt) un.insert(kv.first);
            std::vector<const uint64_t *> as, pts(ng * un.size(), nullptr); std::vector<uint64_t *> outs; std::vector<int> accs(ng, 0);
            size_t t = 0; for (int i : un) { as.push_back(rot[i].get()); for (size_t v = 0; v < ng; v++) if (sums[u + v].pt.count(i)) pts[v * un.size() + t] = sums[u + v].pt[i]; t++; }
            for (size_t v = 0; v < ng; v++) outs.push_back(sums[u + v].out);
            HCR(hc_qp_mul_sum_many(hc, L, (int)as.size(), (int)ng, as.data(), pts.data(), outs.data(), accs.data()));
        }
        for (auto &ix : index) {
            const int j = ix.first; if (j == 0) continue;
            const int g = j * lt.n1; const uint64_t gal = gal_rot(g);
            const auto &row = lt.giant.at(g);
            const bool haveA = Aof.count(j) != 0;
            auto a0 = block(), a1 = block();
            if (haveA) HCR(hc_mod_down2(hc, L, Aof[j].get(), a0.get(), a1.get()));
            else { HCR(hc_lv_mul_const(hc, L, ct.p[0].get(), zeros.data(), a0.get())); HCR(hc_lv_mul_const(hc, L, ct.p[0].get(), zeros.data(), a1.get())); }
            if (row.count(0)) { const uint64_t *pt = row.at(0).p.get(); HCR(hc_lv_op2(hc, HC_LV_MUL_ACC_PLAIN, L, ct.p[0].get(), ct.p[1].get(), pt, nullptr, a0.get(), a1.get(), nullptr)); }
            { auto t = block(); HCR(hc_lv_permute(hc, gal, L, a0.get(), t.get())); add_to_res(0, t); }
            HCR(hc_keyswitch_qp_rotate(hc, key(gal, L, 2), gal, L, nullptr, a1.get(), B.get(), 0, haveB ? 1 : 0)); n_keyswitch++; haveB = true;     // SwitchKeysInPlaceNoModDown, permuted into the accumulators
            Aof.erase(j);
        }
        static const bool split_rescale = getenv("HCONV_NO_FUSED_RESCALE") != nullptr;
        const bool fuse = fuse_min_scale > 0 && haveB && L >= 2 && !split_rescale && res.scale / (double)Q[(size_t)L] >= fuse_min_scale / 2;
        auto diag0 = [&]() {
            if (!(lt.giant.count(0) && lt.giant.at(0).count(0))) return;
            const uint64_t *pt = lt.giant.at(0).at(0).p.get();
            if (have_res[0] && have_res[1]) HCR(hc_lv_op2(hc, HC_LV_MUL_ACC_PLAIN, L, ct.p[0].get(), ct.p[1].get(), pt, nullptr, res.p[0].get(), res.p[1].get(), nullptr));
            else for (int k = 0; k < 2; k++) { auto t = block(); HCR(hc_lv_mul_plain(hc, L, ct.p[k].get(), pt, t.get())); add_to_res(k, t); }
        };
        if (fuse) {            // every other term first (modular sums commute), then ModDown(B) + them + the first drop of the caller's Rescale in one pass
            diag0();
            if (have_res[0] != have_res[1]) for (int k = 0; k < 2; k++) if (!have_res[k]) { auto z = block(); HCR(hc_lv_mul_const(hc, L, ct.p[k].get(), zeros.data(), z.get())); add_to_res(k, z); }
            DCt out = new_ct(L - 1, 1, res.scale / (double)Q[(size_t)L]); alg_ct += 2.0 * (2 * L + 1);
            HCR(hc_mod_down2_add_rescale(hc, L, B.get(), have_res[0] ? res.p[0].get() : nullptr, have_res[0] ? res.p[1].get() : nullptr, out.p[0].get(), out.p[1].get()));
            return out;
        }
        if (haveB) { auto d0 = block(), d1 = block(); HCR(hc_mod_down2(hc, L, B.get(), d0.get(), d1.get())); add_to_res(0, d0); add_to_res(1, d1); }
        diag0();
        if (!have_res[0] || !have_res[1]) panic("linear_transform_qp: empty matrix");
        return res;
    }
    DCt linear_transform(const DCt &ct, const LT &lt, double fuse_min_scale = 0) {             // sum_k diag_k (.) rot_k(ct); no rescale (or its first drop: linear_transform_qp)
        if (ct.level != lt.level) panic("linear_transform: ciphertext level differs from the encoded matrix level");
        return linear_transform_qp(ct, lt, nullptr, -1, fuse_min_scale);
    }

    // ---------------- polynomial evaluation (tests/oracle_ckks.py: _power, _split, _plan_level, _eval_rec, eval_poly)
    struct PolyEval {
        Boot *B; bool cheby; std::map<int, DCt> T;
        const DCt &power(int i) {
            auto it = T.find(i); if (it != T.end()) return it->second;
            const int a = (i + 1) / 2, b = i / 2;
            DCt A = power(a), Bc = power(b);
            DCt t = B->mul_relin_rescale(A, Bc);
            if (cheby) {
                t = B->add(t, t);
                const int c = a - b;
                if (c == 0) t = B->add_const(t, -1.0);
                else { DCt Tc = power(c); const int L = std::min(t.level, Tc.level); t = B->sub(drop_to(t, L), relabel(drop_to(Tc, L), t.scale)); }
            }
            return T.emplace(i, t).first->second;
        }
        static void split(const std::vector<double> &c, int g, bool cheby, std::vector<double> &cq, std::vector<double> &cr) {
            const int deg = (int)c.size() - 1;
            cr.assign(c.begin(), c.begin() + g);
            if (!cheby) { cq.assign(c.begin() + g, c.end()); return; }
            cq.assign((size_t)(deg - g + 1), 0.0); cq[0] = c[(size_t)g];
            for (int j = 1; j <= deg - g; j++) { cq[(size_t)j] = 2.0 * c[(size_t)(g + j)]; cr[(size_t)(g - j)] -= c[(size_t)(g + j)]; }
        }
        static int degree(const std::vector<double> &c) { int d = (int)c.size() - 1; while (d > 0 && c[(size_t)d] == 0) d--; return d; }
        static int bit_length(int x) { int b = 0; while (x) { b++; x >>= 1; } return b; }
        int plan_level(std::vector<double> c, int log_split, bool lead) {
            const int deg = degree(c); c.resize((size_t)deg + 1);
            if (deg < (1 << log_split)) {
                if (lead && log_split > 1 && deg > (1 << (log_split - 1))) return plan_level(c, bit_length(deg) >> 1, true);
                int lv = power(1).level; for (int i = 1; i <= deg; i++) if (c[(size_t)i] != 0) lv = std::min(lv, power(i).level);
                return lv - 1;
            }
            int g = 1 << log_split; while (g * 2 <= deg) g *= 2;
            std::vector<double> cq, cr; split(c, g, cheby, cq, cr);
            const int lq = plan_level(cq, log_split, lead), lr = plan_level(cr, log_split, false);
            return std::min(std::min(lq, power(g).level) - 1, lr);
        }
        DCt rec(std::vector<double> c, int log_split, bool lead, double target) {
            const int deg = degree(c); c.resize((size_t)deg + 1);
            if (deg < (1 << log_split)) {
                if (lead && log_split > 1 && deg > (1 << (log_split - 1))) return rec(c, bit_length(deg) >> 1, true, target);
                int lv = power(1).level; for (int i = 1; i <= deg; i++) if (c[(size_t)i] != 0) lv = std::min(lv, power(i).level);
                const double pre = target * (double)B->Q[(size_t)lv];
                DCt acc; bool have = false;
                for (int i = 1; i <= deg; i++) if (c[(size_t)i] != 0) {
                    DCt Xi = drop_to(power(i), lv);
                    DCt term = B->mul_const_int(Xi, nearbyint(c[(size_t)i] * pre / Xi.scale)); term.scale = pre;
                    acc = have ? B->add(acc, term) : term; have = true;
                }
                if (!have) panic("polynomial leaf without a non-constant term");
                if (c[0] != 0) acc = B->add_const_int(acc, nearbyint(c[0] * pre));
                return relabel(B->rescale(acc), target);
            }
            int g = 1 << log_split; while (g * 2 <= deg) g *= 2;
            std::vector<double> cq, cr; split(c, g, cheby, cq, cr);
            const DCt Xg = power(g);
            const int lq = plan_level(cq, log_split, lead), lmul = std::min(lq, Xg.level);
            DCt resq = rec(cq, log_split, lead, target * (double)B->Q[(size_t)lmul] / Xg.scale);
            if (resq.level != lq) panic("polynomial evaluation: planned level differs");
            DCt prod = relabel(B->mul_relin_rescale(resq, Xg), target);
            bool rnz = false; for (double v : cr) if (v != 0) rnz = true;
            if (rnz) prod = B->add(prod, rec(cr, log_split, false, target));
            return prod;
        }
    };
    // ---------------- ckks.(*evaluator).EvaluatePoly in the standard basis, AS THE REFERENCE'S FORK EVALUATES IT (evalReLU's three sign
    // polynomials, conv.go:460-477): computePowerBasis (C[n] = Rescale(MulRelin(C[ceil n/2], C[n/2]))), recurse / splitCoeffs (baby-step
    // giant-step, the giant power's level picks the modulus that fixes the quotient's target scale), evaluatePolyFromPowerBasis
    // (MultByGaussianIntegerAndAdd with int64(c * targetScale * q / scale_of_power): truncation, not rounding; one Rescale), Add with the
    // smaller-scale operand multiplied by uint64(ratio). tests/lattigo_poly.py is the same code on the oracle and reproduces EVERY nested
    // ciphertext digest the reference binary produced for these polynomials (tests/test_oracle_pin_poly.py, gotrace -poly).
    struct LPoly { std::vector<double> c; int max_deg; bool lead; int degree() const { return (int)c.size() - 1; } };
    DCt lt_rescale(DCt a, double min_scale) { while (a.level > 0 && a.scale / (double)Q[(size_t)a.level] >= min_scale / 2) a = rescale(a); return a; }   // ckks Rescale's drop rule
    DCt mul_relin_lt_rescale(const DCt &a, const DCt &b, double min_scale) {                          // lt_rescale(mul_relin(a, b)), the first drop inside the key switch
        const int L = std::min(a.level, b.level);
        if (L > 0 && a.scale * b.scale / (double)Q[(size_t)L] >= min_scale / 2) return lt_rescale(mul_relin_rescale(a, b), min_scale);
        return mul_relin(a, b);
    }
    DCt lt_add(const DCt &a, const DCt &b) {                           // evaluateInPlace: uint64(ratio) * the smaller-scale operand
        if (a.scale > b.scale) { const double k = floor(a.scale / b.scale); DCt bb = k > 1 ? mul_const_int(b, k) : b; bb.scale = a.scale; return add(a, bb); }
        if (b.scale > a.scale) { const double k = floor(b.scale / a.scale); DCt aa = k > 1 ? mul_const_int(a, k) : a; aa.scale = b.scale; return add(aa, b); }
        return add(a, b);
    }
    // lt_rescale(lt_add(mul_relin(a, b), tmp), sc) with the first drop inside the key switch: tmp (times evaluateInPlace's integer ratio) joins d0, d1 before the relinearisation
    DCt mul_relin_add_lt_rescale(const DCt &a, const DCt &b, const DCt &tmp, double sc) {
        const int L = std::min(a.level, b.level); const double ps = a.scale * b.scale;
        static const bool split = getenv("HCONV_NO_FUSED_RESCALE") != nullptr;
        if (split || L < 2 || tmp.level < L || tmp.deg != 1 || ps < tmp.scale || !(ps / (double)Q[(size_t)L] >= sc / 2)) return lt_rescale(lt_add(mul_relin(a, b), tmp), sc);
        const double k = ps > tmp.scale ? floor(ps / tmp.scale) : 1.0;
        const DCt bb = k > 1 ? mul_const_int(tmp, k) : tmp;
        DCt r = new_ct(L - 1, 1, ps / (double)Q[(size_t)L]); alg_ct += 6.0 * (L + 1) + 6.0 * (L + 1) + 2.0 * (2 * L + 1); alg_shared += ks_rows(L);
        auto d0 = block(), d1 = block(), d2 = block();
        HCR(hc_lv_mul_tensor(hc, L, a.p[0].get(), a.p[1].get(), b.p[0].get(), b.p[1].get(), d0.get(), d1.get(), d2.get()));
        HCR(hc_lv_op2(hc, HC_LV_ADD, L, d0.get(), d1.get(), bb.p[0].get(), bb.p[1].get(), d0.get(), d1.get(), nullptr));
        HCR(hc_keyswitch_add_rescale(hc, key(0, L), L, d2.get(), d0.get(), d1.get(), r.p[0].get(), r.p[1].get())); n_keyswitch++;
        return lt_rescale(r, sc);
    }
    void lt_power(std::map<int, DCt> &C, int n, double sc) {
        if (C.count(n)) return;
        const int a = (n + 1) / 2, b = n >> 1;
        lt_power(C, a, sc); lt_power(C, b, sc);
        C[n] = mul_relin_lt_rescale(C[a], C[b], sc);
    }
    DCt lt_leaf(double target, const LPoly &p, std::map<int, DCt> &C, double sc) {
        if (p.degree() == 0) panic("EvaluatePoly: constant leaf (not produced by the sign polynomials)");
        const int lv = C[p.degree()].level; const double qi = (double)Q[(size_t)lv];
        if (fabs(p.c[0]) > 1e-14) panic("EvaluatePoly: constant term in a leaf (AddConst; not produced by the sign polynomials)");
        std::vector<DCt> as; std::vector<double> ks;
        for (int key = p.degree(); key > 0; key--) if (fabs(p.c[(size_t)key]) > 1e-14) {
            const double const_scale = target * qi / C[key].scale;
            as.push_back(C[key]); ks.push_back(trunc(p.c[(size_t)key] * const_scale));                     // Go's int64(float64)
        }
        if (as.empty()) panic("EvaluatePoly: empty leaf");
        return lt_rescale(lincomb(as, ks, lv, target * qi), sc);                                           // MultByGaussianIntegerAndAdd per power: one launch
    }
    DCt lt_recurse(double target, int log_split, int log_degree, const LPoly &p, std::map<int, DCt> &C, double sc) {
        if (p.degree() < (1 << log_split)) {
            if (p.lead && log_split > 1 && p.degree() > (1 << (log_split - 1))) { const int ld = PolyEval::bit_length(p.degree()); return lt_recurse(target, ld >> 1, ld, p, C, sc); }
            return lt_leaf(target, p, C, sc);
        }
        int next_power = 1 << log_split; while (next_power < (p.degree() >> 1) + 1) next_power <<= 1;
        LPoly pr{std::vector<double>(p.c.begin(), p.c.begin() + next_power), p.max_deg == p.degree() ? next_power - 1 : p.max_deg - (p.degree() - next_power + 1), false};
        LPoly pq{std::vector<double>(p.c.begin() + next_power, p.c.end()), p.max_deg, p.lead};
        int level = C[next_power].level - 1; if (p.max_deg >= (1 << (log_degree - 1)) && p.lead) level++;
        DCt res = lt_recurse(target * (double)Q[(size_t)level] / C[next_power].scale, log_split, log_degree, pq, C, sc);
        DCt tmp = lt_recurse(target, log_split, log_degree, pr, C, sc);
        if (res.level > tmp.level) res = drop_to(res, tmp.level + 1);                                     // DropLevel
        if (std::min(res.level, C[next_power].level) > tmp.level) { res = mul_relin_lt_rescale(res, C[next_power], sc); res = lt_add(res, tmp); }
        else res = mul_relin_add_lt_rescale(res, C[next_power], tmp, sc);
        return res;
    }
    // ---- the Chebyshev basis of the same evaluator (EvaluateCheby @52d7c0: computePowerBasisCheby, splitCoeffsCheby, recurseCheby; the leaf is
    // shared): tests/lattigo_poly.py's Chebyshev path, which reproduces every nested digest of the binary's sine evaluation
    // (tests/test_oracle_pin_cheby.py; on the GPU test_evaluate_cheby_vs_reference_trace_on_gpu)
    DCt lt_sub(const DCt &a, const DCt &b) {
        if (a.scale > b.scale) { const double k = floor(a.scale / b.scale); DCt bb = k > 1 ? mul_const_int(b, k) : b; bb.scale = a.scale; return sub(a, bb); }
        if (b.scale > a.scale) { const double k = floor(b.scale / a.scale); DCt aa = k > 1 ? mul_const_int(a, k) : a; aa.scale = b.scale; return sub(aa, b); }
        return sub(a, b);
    }
    void lt_power_cheby(std::map<int, DCt> &C, int n, double sc) {                   // C[n] = 2 C[a] C[b] - C[a-b]
        if (C.count(n)) return;
        const int a = (n + 1) / 2, b = n >> 1, c = a - b;
        lt_power_cheby(C, a, sc); lt_power_cheby(C, b, sc); if (c) lt_power_cheby(C, c, sc);
        DCt t = mul_relin_lt_rescale(C[a], C[b], sc);
        // 2 t - 1 (AddConst: floor(|scale| + 0.5)) or 2 t - C[c] (evaluateInPlace's Sub: the smaller-scale operand times uint64(ratio)) as ONE launch: the residues of Add(t, t)
        // followed by AddConst / Sub, without their passes over the ciphertext
        if (c == 0) C[n] = lincomb({t}, {2.0}, t.level, t.scale, true, -floor(fabs(t.scale) + 0.5));
        else {
            const DCt &u = C[c]; const int L = std::min(t.level, u.level);
            double kt = 1, ku = 1, scale = t.scale;
            if (t.scale > u.scale) ku = std::max(1.0, floor(t.scale / u.scale)); else if (u.scale > t.scale) { kt = std::max(1.0, floor(u.scale / t.scale)); scale = u.scale; }
            C[n] = lincomb({t, u}, {2.0 * kt, -ku}, L, scale);
        }
    }
    DCt lt_leaf_any(double target, const LPoly &p, std::map<int, DCt> &C, double sc) {       // evaluatePolyFromPowerBasis incl. the constant term
        const bool c0 = fabs(p.c[0]) > 1e-14;
        if (p.degree() == 0) { DCt z = mul_const_int(C[1], 0.0); z.scale = target; return c0 ? add_const(z, p.c[0]) : z; }
        const int lv = C[p.degree()].level; const double qi = (double)Q[(size_t)lv];
        std::vector<DCt> as; std::vector<double> ks;
        for (int key = p.degree(); key > 0; key--) if (fabs(p.c[(size_t)key]) > 1e-14) {
            const double const_scale = target * qi / C[key].scale;
            as.push_back(C[key]); ks.push_back(trunc(p.c[(size_t)key] * const_scale));                     // Go's int64(float64)
        }
        if (as.empty()) { as.push_back(C[1]); ks.push_back(0.0); }
        // the constant term (evaluator.AddConst: floor(|c * scale| + 0.5), sign restored) rides in the same launch; the reference adds it first, residues mod q do not depend on the order
        const double k0 = floor(fabs(target * qi * p.c[0]) + 0.5) * (p.c[0] < 0 ? -1.0 : 1.0);
        return lt_rescale(lincomb(as, ks, lv, target * qi, c0, k0), sc);
    }
    DCt lt_recurse_cheby(double target, int log_split, int log_degree, const LPoly &p, std::map<int, DCt> &C, double sc) {
        if (p.degree() < (1 << log_split)) {
            if (p.lead && log_split > 1 && p.max_deg > ((1 << log_degree) - (1 << (log_split - 1)))) { const int ld = PolyEval::bit_length(p.degree()); return lt_recurse_cheby(target, ld >> 1, ld, p, C, sc); }
            return lt_leaf_any(target, p, C, sc);
        }
        int next_power = 1 << log_split; while (next_power < (p.degree() >> 1) + 1) next_power <<= 1;
        LPoly pr{std::vector<double>(p.c.begin(), p.c.begin() + next_power), p.max_deg == p.degree() ? next_power - 1 : p.max_deg - (p.degree() - next_power + 1), false};
        LPoly pq{std::vector<double>((size_t)(p.degree() - next_power + 1), 0.0), p.max_deg, p.lead};
        pq.c[0] = p.c[(size_t)next_power];
        for (int i = next_power + 1, j = 1; i <= p.degree(); i++, j++) { pq.c[(size_t)(i - next_power)] = 2 * p.c[(size_t)i]; pr.c[(size_t)(next_power - j)] -= p.c[(size_t)i]; }
        int level = C[next_power].level - 1; if (p.max_deg >= (1 << (log_degree - 1)) && p.lead) level++;
        DCt res = lt_recurse_cheby(target * (double)Q[(size_t)level] / C[next_power].scale, log_split, log_degree, pq, C, sc);
        DCt tmp = lt_recurse_cheby(target, log_split, log_degree, pr, C, sc);
        if (res.level > tmp.level) res = drop_to(res, tmp.level + 1);
        if (std::min(res.level, C[next_power].level) > tmp.level) { res = mul_relin_lt_rescale(res, C[next_power], sc); res = lt_add(res, tmp); }
        else res = mul_relin_add_lt_rescale(res, C[next_power], tmp, sc);
        return res;
    }
    DCt eval_cheby_lattigo(const DCt &ct, const std::vector<double> &coeffs, double target, double sc) {
        std::map<int, DCt> C; C[1] = ct;
        LPoly p{coeffs, (int)coeffs.size() - 1, true};
        const int log_degree = PolyEval::bit_length(p.degree()), log_split = log_degree >> 1;
        for (int i = 2; i < (1 << log_split); i++) lt_power_cheby(C, i, sc);
        for (int i = log_split; i < log_degree; i++) lt_power_cheby(C, 1 << i, sc);
        return lt_recurse_cheby(target, log_split, log_degree, p, C, sc);
    }
    DCt eval_poly_lattigo(const DCt &ct, const std::vector<double> &coeffs, double target) {
        std::map<int, DCt> C; C[1] = ct;
        LPoly p{coeffs, (int)coeffs.size() - 1, true};
        const double sc = 1073741824.0;                                                                  // evaluator.scale = params.Scale()
        const int log_degree = PolyEval::bit_length(p.degree()), log_split = log_degree >> 1;
        for (int i = 2; i < (1 << log_split); i++) lt_power(C, i, sc);
        for (int i = log_split; i < log_degree; i++) lt_power(C, 1 << i, sc);
        return lt_recurse(target, log_split, log_degree, p, C, sc);
    }
    DCt eval_poly(const DCt &ct, const std::vector<double> &coeffs, double target, bool cheby) {
        if (!cheby) return eval_poly_lattigo(ct, coeffs, target);
        PolyEval pe{this, cheby, {}};
        pe.T[1] = ct;
        const int deg = (int)coeffs.size() - 1, log_deg = PolyEval::bit_length(deg), log_split = log_deg >> 1;
        for (int i = 2; i < (1 << log_split); i++) pe.power(i);
        for (int i = log_split; i < log_deg; i++) pe.power(1 << i);
        return pe.rec(coeffs, log_split, true, target);
    }

    // ---------------- the bootstrapper
    void build(const std::vector<int64_t> &sk_in, const Seed256 &seed, int device, int chain_ = 6) {
        chain = chain_;
        if (chain == 7 && testOnlyEnv("HCONV_CHAIN_REPLAY_BL")) { replay_seed = strtoull(testOnlyEnv("HCONV_CHAIN_REPLAY_BL"), nullptr, 0); fprintf(stderr, "hconv: HCONV_CHAIN_REPLAY_BL: planted keys and input for the baseline's Bootstrapp (test mode; the run ends behind it)\n"); }
        if (chain == 6 && testOnlyEnv("HCONV_CHAIN_REPLAY")) { replay_seed = strtoull(testOnlyEnv("HCONV_CHAIN_REPLAY"), nullptr, 0); fprintf(stderr, "hconv: HCONV_CHAIN_REPLAY: planted keys and input (test mode; results are meaningless as ciphertexts)\n"); }
        Q = chain == 7 ? PARAMS7_Q : PARAMS6_Q; P = PARAMS6_P; NQ = (int)Q.size(); sk = sk_in; rng.reseed(seed, 0xB007B007ull + (uint64_t)chain_);
        // parameter set [7] (kind "BL_Conv", main.go:52-55): the stock NewBootstrapper (main.go:476-479). SlotsToCoeffs sits right behind the sine on levels 15, 15, 14 at
        // plaintext scales sqrt(q15), sqrt(q15), 2^30 (tests/golden/ref_flow_bl_5_1.json)
        if (chain == 7) { LV_STC_TOP = 15; stc_scale_top = sqrt((double)Q[15]); stc_scale_last = 1073741824.0; lv_relin_lo = 2; }
        else { LV_STC_TOP = 3; stc_scale_top = sqrt((double)Q[3]); stc_scale_last = 1073741824.0; lv_relin_lo = LV_RELU_TOP - 11; }
        if (hc_ctx_create(&hc, LOGN, Q.data(), NQ, P.data(), (int)P.size(), device)) panic(std::string("hc_ctx_create: ") + hc_last_error(nullptr));
        applyEnvOptions(hc, 2);                                                        // pack32 = 2 unless HCONV_PACK32 says otherwise: 4-byte rows for the ~30-bit limbs of every leveled operand (HCONV_PACK32=0 / 1: A/B against the 8-byte forms)
        row32.assign((size_t)NQ, 0); for (int l = 0; l < NQ; l++) row32[(size_t)l] = (char)hc_row_is32(hc, l);
        const int nm = NQ + (int)P.size();
        { void *v = nullptr; HCR(hc_malloc(hc, (size_t)nm * N * 8, &v)); d_sk = (uint64_t *)v; }
        std::vector<uint64_t> h((size_t)N);
        for (int m = 0; m < nm; m++) {
            const uint64_t q = m < NQ ? Q[(size_t)m] : P[(size_t)(m - NQ)];
            for (int j = 0; j < N; j++) h[(size_t)j] = sk[(size_t)j] >= 0 ? (uint64_t)sk[(size_t)j] : q - (uint64_t)(-sk[(size_t)j]);
            HCR(hc_upload(hc, d_sk + (size_t)m * N, h.data(), (size_t)N * 8)); HCR(hc_ntt(hc, m, d_sk + (size_t)m * N, d_sk + (size_t)m * N, 1));
        }
        mono_i = block1();
        { std::vector<uint64_t> m((size_t)NQ * N, 0); for (int l = 0; l < NQ; l++) m[(size_t)l * N + N / 2] = 1; pack_rows(m, NQ); HCR(hc_upload(hc, mono_i.get(), m.data(), m.size() * 8)); HCR(hc_lv_ntt(hc, NQ - 1, mono_i.get(), mono_i.get())); }
        // Chebyshev interpolant of cos(2 pi (K u - 1/4) / 2^r) on [-1,1]
        const int m = SIN_DEG + 1; sine.assign((size_t)m, 0.0);
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) { const double u = cos(M_PI * (k + 0.5) / m); s += cos(2.0 * M_PI * (SIN_K * u - 0.25) / (double)(1 << SIN_DOUBLE)) * cos(j * M_PI * (k + 0.5) / m); }
            sine[(size_t)j] = 2.0 / m * s;
        }
        sine[0] /= 2;
        key(2ull * N - 1, LV_SINE_TOP);                                      // conjugation
        for (int l = LV_SINE_TOP; l >= lv_relin_lo; l--) if (chain == 6 || l > LV_RELU_TOP || l <= 12) key(0, l);     // kgen.GenRelinearizationKey (main.go:411), at the levels that multiply
        if (chain == 7) { key(2ull * N - 1, 1); key(2ull * N - 1, 12); }       // pack_evaluator.ConjugateNew before / after Bootstrapp (test_BL.go:116, 155)
    }
    // One bootstrapper (main.go:480-507: btp for log_sparse 0, btp2..btp5 for 1..4): its DFT matrices, encoded, and every rotation
    // key it switches with (GenRotationKeysForRotations(btpParams.RotationsForBootstrapping(LogSlots)), main.go:466-474).
    // log_sparse = ls > 0 (sparse packing, coefficients on the multiples of D = 2^ls): the subring X^D with n_s = n/D slots; BOTH
    // coefficient halves travel in ONE ciphertext (first half of every 2 n_s slots = low half, second = high half).
    Set &set(int ls) {
        auto it = sets.find(ls); if (it != sets.end()) return it->second;
        Set S; S.ls = ls; S.ns = n >> ls;
        if (!dft_digests && getenv("HCONV_DFT_DIGESTS") && *getenv("HCONV_DFT_DIGESTS")) { dft_digests = fopen(getenv("HCONV_DFT_DIGESTS"), "a"); if (!dft_digests) panic("HCONV_DFT_DIGESTS: cannot open the file"); }
        const int D = 1 << ls, ns = S.ns;
        // CoeffsToSlots: (1/n_s) prod(stages), times 1/2 (real/imaginary extraction), 1/K (Chebyshev argument in [-1,1]), 1/D (SubSum)
        // Full slots: the fork's own matrices (lattigo_dft) with its constant coeffsToSlotsDiffScale = (2 / ((b-a) N scFac qDiff))^(1/4), b-a = 2K/scFac,
        // qDiff = q0 / 2^round(log2 q0) (genDFTMatrices) - the same 1/(2 n K) as below divided by qDiff, which ctos() accounts for by
        // labelling the raised ciphertext 2^round(log2 q0) instead of q0 - and its baby-step size N1. For parameter set [6] these are the
        // reference binary's diagonals bit for bit (tests/golden/ref_trace_diag_5_1.json). Sparse slots keep this file's own generator.
        const double scfac = (double)(1 << SIN_DOUBLE), qdiff = (double)Q[0] / exp2(round(log2((double)Q[0])));
        const bool fork = chain == 6;           // parameter set [6] (Ours): the fork's matrices, sparse slots included (round 3); the baseline's chain keeps this file's generator for ls > 0
        const int period = ls && fork ? 2 * ns : 0;
        std::vector<DiagMat> G = ls && !fork ? dft_groups(true, {4, 4, 4, 3}, 1.0 / (2.0 * (double)ns * SIN_K * D), ls)
                                             : lattigo_dft(true, 4, pow(2.0 / ((2.0 * SIN_K / scfac) * (double)N * scfac * qdiff), 1.0 / 4.0), ls);
        if (ls && !fork) for (auto &e : G.back()) for (int p = 0; p < n; p++) if (p % (2 * ns) >= ns) e.second[(size_t)p] = cplx(0, 0);   // keep w on the first half of every 2 n_s slots
        static const char *cts_tag[4] = {"cts0", "cts1", "cts2", "cts3"}, *stc_tag[3] = {"stc0", "stc1", "stc2"};
        for (size_t i = 0; i < G.size(); i++) { const int lv = LV_CTS_TOP - (int)i; S.cts.push_back(plan(G[i], lv, (double)Q[(size_t)lv], ls == 0 || fork, cts_tag[i], period)); }
        // SlotsToCoeffs: level 3 carries all but the last matrix (plaintext scales multiply to q3), level 2 the last at 2^30
        // (the fork: the set NewBootstrapper_mod builds with scale 1 - the reference's SlotsToCoeffs call uses it - at the same three scales)
        // (the stock Bootstrapp of parameter set [7]: the set with the bootstrapping scale, (qDiff * params.scale / prescale)^(1/3) per matrix - matrices 4-6 of ref_trace_diag_5_1.json)
        const double stc_const = chain == 7 ? pow(qdiff * 1073741824.0 / exp2(round(log2((double)Q[0] / 256.0))), 1.0 / 3.0) : 1.0;
        G = ls && !fork ? dft_groups(false, {5, 5, 5}, 1.0, ls) : lattigo_dft(false, 3, stc_const, ls);
        if (ls && !fork) {       // packed a = (re | im)  ->  w = re + i im on both halves:  w = (m1 + i m2) a + (i m1 + m2) rot_{n_s}(a)
            DiagMat W; W[0].resize((size_t)n); W[ns].resize((size_t)n);
            for (int p = 0; p < n; p++) { const bool first = p % (2 * ns) < ns; W[0][(size_t)p] = first ? cplx(1, 0) : cplx(0, 1); W[ns][(size_t)p] = first ? cplx(0, 1) : cplx(1, 0); }
            G[0] = matmul_diag(G[0], W, 2 * ns);
        }
        if (G.size() != 3) panic("SlotsToCoeffs is planned as three matrices");
        for (size_t i = 0; i + 1 < G.size(); i++) S.stc.push_back(plan(G[i], LV_STC_TOP, stc_scale_top, ls == 0 || fork, stc_tag[i], period));
        // the fork applies all three matrices on level 3 and rescales twice afterwards (ckks.SlotsToCoeffs -> dft: the Rescale after each
        // LinearTransform finds nothing to drop); full slots on parameter set [6] do the same, the other bootstrappers keep the split above
        S.stc.push_back(plan(G.back(), fork ? LV_STC_TOP : LV_STC_TOP - 1, stc_scale_last, ls == 0 || fork, stc_tag[2], period));
        for (auto *grp : {&S.cts, &S.stc}) for (auto &lt : *grp) for (auto &g : lt.giant) {
            if (g.first) key(gal_rot(g.first), lt.level, 2);
            for (auto &b : g.second) if (b.first) key(gal_rot(b.first), lt.level, 1);
        }
        // a SlotsToCoeffs matrix below its predecessor's level (the stock Bootstrapp's last one) hoists its baby steps at the ciphertext's level (linear_transform_qp, hoist_c1)
        for (size_t i = 1; i < S.stc.size(); i++) if (S.stc[i].level < S.stc[i - 1].level)
            for (auto &g : S.stc[i].giant) for (auto &b : g.second) if (b.first) key(gal_rot(b.first), S.stc[i - 1].level, 1);
        for (int j = 0; j < ls; j++) key(gal_rot(ns << j), LV_CTS_TOP);       // SubSum
        if (ls) key(gal_rot(ns), LV_SINE_TOP);                                 // packing the imaginary half next to the real one
        if (dft_digests) fflush(dft_digests);
        return sets.emplace(ls, std::move(S)).first->second;
    }
    // level-0 coefficient-encoded ciphertext -> slot-encoded at level 15, scale 2^30: two ciphertexts (low / high coefficient
    // half, bit-reversed order) for ls = 0, one packed ciphertext for ls > 0
    // Full slots on parameter set [6]: ckks.(*Bootstrapper).BootstrappConv_CtoS op for op as the reference's fork runs it (gotrace -flow,
    // tests/golden/ref_flow_5_1.json; tests/oracle_ckks.py Bootstrapper._ctos_fork is the same on the oracle): ScaleUp to prescale =
    // 2^round(log2(q0 / MessageRatio)), modUp, ScaleUp to sinescale / MessageRatio, four times LinearTransform + Rescale(min = the scale
    // before), ct + conj and (ct - conj) / i, the label sinescale = 2^round(log2 q0), AddConst(-0.5 / (scFac (b - a))), EvaluateCheby with the
    // fork's coefficients towards sqrt(sqrt(sinescale q16) q17), two double angles with (1/2pi)^(1/4) squared along, the label params.scale,
    // MultByConst(q0 / sinescale * params.scale / prescale) and Rescale: two ciphertexts at level 14, scale 2^30.
    // Sparse slots (ls > 0, round 3; gotrace -flow -logslots 13 shows the binary's op sequence): subSum after the second ScaleUp (Rotate by 2^i,
    // Add, i = logSlots .. logN-2), and after DivByi the repacking Rotate(ct1, 2^logSlots) + Add(ct0, ct1): one ciphertext through the sine.
    // stock = true: the first half of the stock ckks.(*Bootstrapper).Bootstrapp (the baseline, parameter set [7]; gotrace -flow-bl, tests/golden/ref_flow_bl_5_1.json): SetScale(ct,
    // prescale) - MultByConst + Rescale down to level 0 - instead of the first ScaleUp, the same modUp / ScaleUp / CoeffsToSlots / evaluateSine, and no MultByConst + Rescale at
    // the end: two ciphertexts at level 15, scale 2^30, for SlotsToCoeffs.
    int ctos_fork(const DCt &ct0, DCt out[2], int ls = 0, bool stock = false) {
        Set &S = set(ls);
        const double q0 = (double)Q[0], msg_ratio = 256.0, pscale = 1073741824.0;
        const double prescale = exp2(round(log2(q0 / msg_ratio))), sinescale = exp2(round(log2(q0)));
        DCt ct;
        if (stock) { ct = set_scale(ct0, prescale); if (ct.level != 0) panic("Bootstrapp: SetScale did not reach level 0"); }
        else {
            if (ct0.level != 0 || prescale < ct0.scale) panic("BootstrappConv_CtoS: the input must sit on level 0 below the prescale");
            const double k0 = floor(prescale / ct0.scale + 0.5);
            ct = mul_const_int(ct0, k0); ct.scale = ct0.scale * k0;
        }
        double k;
        ct = mod_raise(ct, LV_CTS_TOP);
        k = floor((sinescale / msg_ratio) / ct.scale + 0.5);
        { const double s0 = ct.scale; ct = mul_const_int(ct, k); ct.scale = s0 * k; }
        for (int i = LOGN - 1 - ls; i < LOGN - 1; i++) ct = add(ct, rotate(ct, 1 << i));                   // subSum
        for (auto &lt : S.cts) { const double s_in = ct.scale; ct = lt_rescale(linear_transform(ct, lt, s_in), s_in); }
        if (ct.level != LV_SINE_TOP) panic("CoeffsToSlots ended at the wrong level");
        if (profiling()) profile_dump(this, "modUp + CoeffsToSlots");
        DCt cc = conjugate(ct);
        DCt parts[2] = {add(ct, cc), mul_by_i(sub(cc, ct))};           // DivByi(ct - conj) = -i (ct - conj) = i (conj - ct): the same residues
        const int nparts = ls ? 1 : 2;
        if (ls) parts[0] = add(parts[0], rotate(parts[1], S.ns));      // repacking: the imaginary half next to the real one
        double target = sinescale;
        for (int r = 0; r < SIN_DOUBLE; r++) target = sqrt(target * (double)Q[(size_t)(LV_RELU_TOP + 1 + r)]);
        const std::vector<double> coeffs(FORK_SINE_COEFFS, FORK_SINE_COEFFS + 63);
        const double scfac = (double)(1 << SIN_DOUBLE);
        const bool merged = nparts == 2 && merge_parts && 2 * nb <= nb_max && !stock;
        if (merged) { parts[0] = merge2(parts[0], parts[1]); }        // both halves through the sine as one batch; split again by the caller after the ReLU (parts_merged)
        parts_merged = merged;
        for (int h = 0; h < (merged ? 1 : nparts); h++) {
            DCt c = parts[h]; c.scale = sinescale;
            c = add_const(c, -0.5 / (scfac * (2.0 * SIN_K / scfac)));
            c = eval_cheby_lattigo(c, coeffs, target, sinescale);
            double sqrt2pi = pow(0.15915494309189535, 1.0 / scfac);
            for (int r = 0; r < SIN_DOUBLE; r++) { sqrt2pi *= sqrt2pi; c = mul_relin(c, c); c = add(c, c); c = lt_rescale(add_const(c, -sqrt2pi), sinescale); }
            if (c.level != LV_RELU_TOP) panic("sine evaluation ended at the wrong level");
            c.scale = pscale;
            out[h] = stock ? c : lt_rescale(mul_const_float(c, (q0 / sinescale) * (pscale / prescale)), pscale);
        }
        return nparts;
    }
    int ctos(const DCt &ct0, int ls, DCt out[2]) {
        if (chain == 6) return ctos_fork(ct0, out, ls);
        if (chain == 7 && ls == 0) return ctos_fork(ct0, out, 0, true);
        Set &S = set(ls);
        const double q0 = (double)Q[0], msg_scale = ct0.scale;
        DCt ct = mod_raise(ct0, LV_CTS_TOP); ct.scale = ls ? q0 : exp2(round(log2(q0)));   // slot values are now t'/Q0 = I + msg/Q0, |.| <= K (full slots: the 1/qDiff sits in the matrices)
        for (int j = 0; j < ls; j++) ct = add(ct, rotate(ct, S.ns << j));                                  // SubSum: trace onto X^D
        for (auto &lt : S.cts) ct = rescale(linear_transform(ct, lt));
        if (ct.level != LV_SINE_TOP) panic("CoeffsToSlots ended at the wrong level");
        DCt cc = conjugate(ct);
        DCt parts[2] = {add(ct, cc), mul_by_i(sub(cc, ct))};           // (w + conj w), -i (w - conj w); the 1/2 is in the matrices
        int np = 2;
        if (ls) { parts[0] = add(parts[0], rotate(parts[1], S.ns)); np = 1; }
        const double c_m = q0 / (2.0 * M_PI * msg_scale);
        double s = sine_out_scale * c_m;
        for (int r = 0; r < SIN_DOUBLE; r++) s = sqrt(s * (double)Q[(size_t)(LV_RELU_TOP + 1 + r)]);
        for (int h = 0; h < np; h++) {
            DCt c = eval_poly(parts[h], sine, s, true);
            for (int r = 0; r < SIN_DOUBLE; r++) { c = mul_relin(c, c); c = add(c, c); c = rescale(add_const(c, -1.0)); }
            if (c.level != LV_RELU_TOP) panic("sine evaluation ended at the wrong level");
            c.scale = c.scale / c_m;                                     // value *= c_m: now msg / msg_scale
            out[h] = c;
        }
        return np;
    }
    DCt stoc(const DCt &re, const DCt *im, int ls) {
        Set &S = set(ls);
        if (ls && im) panic("sparse SlotsToCoeffs takes one packed ciphertext");
        if (chain == 7 && ls == 0) {      // ckks.SlotsToCoeffs inside the stock Bootstrapp (ref_flow_bl_5_1.json): MultByi + Add, LinearTransform on the matrices' own levels 15, 15, 14, each followed by a Rescale(min = the scale before) that finds nothing to drop: level 14, scale ~2^120
            DCt ct = add(re, mul_by_i(*im));
            std::shared_ptr<uint64_t> prev_c1;                                       // the second polynomial of the previous LinearTransform's input (see linear_transform_qp)
            for (auto &lt : S.stc) {
                const double s_in = ct.scale;
                if (ct.level > lt.level) {
                    if (!prev_c1) panic("SlotsToCoeffs: a matrix below the ciphertext's level must follow one at that level");
                    const int Lh = ct.level, nk = lt.level + 1;
                    auto y = block();                                                                            // limbs 0..matrix level of c1, the previous input's limbs above
                    copy_rows(y.get(), ct.p[1].get(), (size_t)nk);
                    copy_rows(y.get(), prev_c1.get(), (size_t)(Lh + 1 - nk), (size_t)nk, (size_t)nk);
                    ct = lt_rescale(linear_transform_qp(drop_to(ct, lt.level), lt, y.get(), Lh), s_in);
                } else { prev_c1 = ct.p[1]; ct = lt_rescale(linear_transform(ct, lt, s_in), s_in); }
            }
            return ct;
        }
        DCt ct = drop_to(ls ? re : add(re, mul_by_i(*im)), LV_STC_TOP);
        if (chain == 6) {        // ckks.SlotsToCoeffs as the fork runs it (sparse slots: one packed ciphertext, no MultByi + Add) (ref_flow_5_1.json): MultByi + Add, three LinearTransforms each followed by Rescale(min = the scale before), then eval.go:564's Rescale(2^30): level 3 -> 1
            for (auto &lt : S.stc) { const double s_in = ct.scale; ct = lt_rescale(linear_transform(ct, lt, s_in), s_in); }
            return lt_rescale(ct, 1073741824.0);
        }
        for (size_t i = 0; i + 1 < S.stc.size(); i++) ct = linear_transform(ct, S.stc[i]);
        ct = rescale(ct);
        return rescale(linear_transform(ct, S.stc.back()));
    }
};
// rot_util.go:141-174
static std::vector<int> gen_keep_vec(int vec_size, int in_wid, int kp_wid, int ul) {
    int logN = 0; for (; (1 << logN) < 2 * vec_size; logN++) {}
    std::vector<int> idx((size_t)vec_size, 0); const int batch = 2 * vec_size / (in_wid * in_wid);
    if (kp_wid < in_wid / 2) panic("keep width too small. less than in_wid/2");
    if (ul != 0 && ul != 1) panic("ul not 0 nor 1");
    const int rows = ul == 0 ? in_wid / 2 : kp_wid - in_wid / 2;
    for (int i = 0; i < rows; i++) for (int j = 0; j < kp_wid; j++) for (int b = 0; b < batch; b++) {
        uint32_t v = (uint32_t)(in_wid * batch * i + batch * j + b), r = 0; for (int k = 0; k < logN - 1; k++) r |= ((v >> k) & 1u) << (logN - 2 - k);
        idx[r] = 1;
    }
    return idx;
}
// rot_util.go:179-218: one mask for the packed (low | high) ciphertext of sparse bootstrapping, period 2 n_s
static std::vector<int> gen_keep_vec_sparse(int vec_size, int in_wid, int kp_wid, int log_sparse) {
    int logN = 0; for (; (1 << logN) < 2 * vec_size; logN++) {}
    std::vector<int> idx((size_t)vec_size, 0); const int batch = 2 * vec_size / (in_wid * in_wid), sparsity = 1 << log_sparse;
    if (sparsity == 1) panic("We do not support full packing in gen_keep_vec_sparse");
    if (kp_wid < in_wid / 2) panic("keep width too small. less than in_wid/2");
    auto rev = [&](int x) { uint32_t v = (uint32_t)x, r = 0; for (int k = 0; k < logN - 1; k++) r |= ((v >> k) & 1u) << (logN - 2 - k); return (int)r; };
    for (int i = 0; i < in_wid / 2; i++) for (int j = 0; j < kp_wid; j++) for (int b = 0; b < batch / sparsity; b++) idx[(size_t)rev(in_wid * batch * i + batch * j + b * sparsity)] = 1;
    for (int i = 0; i < kp_wid - in_wid / 2; i++) for (int j = 0; j < kp_wid; j++) for (int b = 0; b < batch / sparsity; b++) idx[(size_t)(rev(in_wid * batch * i + batch * j + b * sparsity) + vec_size / sparsity)] = 1;
    const int post_slot = 2 * vec_size / sparsity;
    for (int i = 0; i < post_slot; i++) for (int j = 1; j < sparsity / 2; j++) idx[(size_t)(i + post_slot * j)] = idx[(size_t)i];
    return idx;
}
// rot_util.go:557-722: masks and rotations of the two stages of ext_double_ctxt that keep the stride-2 positions and re-pack them
// for the next (half-width) block. log_sparse != 0: one packed ciphertext (ul unused). log_sparse == 0 (full packing, the wide
// networks' first stride layer): the upper (ul = 0) and lower (ul = 1) ciphertexts get their own masks; pos = 0 only.
typedef std::map<int, std::vector<int>> IdxMap;
static void gen_comprs_sparse(int vec_size, int in_wid, int kp_wid, int log_sparse, int ul, IdxMap &m_idx, IdxMap &r_idx) {
    if (in_wid % 2) panic("input wid not divisible by 2");
    const int batch = 2 * vec_size / (in_wid * in_wid * (1 << log_sparse)), min_wid = in_wid / 2;
    int log_in_wid = 0; for (; (1 << log_in_wid) < in_wid; log_in_wid++) {}
    auto rev = [](int x, int bits) { int r = 0; for (int k = 0; k < bits; k++) r |= ((x >> k) & 1) << (bits - 1 - k); return r; };
    if (log_sparse != 0) {
        const int rep = 1 << (log_sparse - 1);
        auto tile = [&](std::vector<int> &t) { const int seg = vec_size / rep; for (int i = 0; i < seg; i++) for (int k = 1; k < rep; k++) t[(size_t)(i + k * seg)] = t[(size_t)i]; };
        for (int j = 0; j < min_wid; j++) {
            std::vector<int> tmp((size_t)vec_size, 0);
            for (int b = 0; b < batch; b++) for (int i = 0; i < min_wid / 2; i++) for (int k = 0; k < 2; k++)
                if (rev(j, log_in_wid - 1) < kp_wid && rev(i, log_in_wid - 2) + k * min_wid / 2 < kp_wid) tmp[(size_t)(k * in_wid * min_wid * batch + in_wid * in_wid * b / 2 + in_wid * j / 2 + i)] = 1;
            tile(tmp); m_idx[j * min_wid / 2] = tmp;
        }
        for (int b = 0; b < batch; b++) {
            std::vector<int> tmp((size_t)vec_size, 0);
            for (int j = 0; j < min_wid; j++) for (int i = 0; i < min_wid / 2; i++) for (int k = 0; k < 2; k++) tmp[(size_t)(k * in_wid * min_wid * batch + b * in_wid * in_wid / 2 + j * min_wid / 2 + i)] = 1;
            tile(tmp); r_idx[3 * b * min_wid * min_wid / 2] = tmp;
        }
        return;
    }
    if (ul != 0 && ul != 1) panic("ul not 0 nor 1");
    auto keep = [&](int j, int i) { return rev(j, log_in_wid - 1) < kp_wid && rev(i, log_in_wid - 2) + (ul ? min_wid / 2 : 0) < kp_wid; };
    const int G = batch > 8 * min_wid ? 8 : (batch > 4 * min_wid ? 4 : 1);          // rot_util.go:617, 656, 693: blocks of G batches move together
    for (int j = 0; j < min_wid; j++) for (int bk = 0; bk < G; bk++) {
        std::vector<int> tmp((size_t)vec_size, 0);
        for (int b = 0; b < batch / G; b++) for (int i = 0; i < min_wid / 2; i++) if (keep(j, i)) tmp[(size_t)(G * in_wid * min_wid * b + bk * min_wid * in_wid + min_wid * j + i)] = 1;
        m_idx[j * min_wid / 2 + (G - 1) * bk * min_wid * min_wid / 2] = tmp;
    }
    for (int b = 0; b < batch / G; b++) {
        std::vector<int> tmp((size_t)vec_size, 0);
        for (int bk = 0; bk < G; bk++) for (int j = 0; j < min_wid; j++) for (int i = 0; i < min_wid / 2; i++) tmp[(size_t)(G * b * in_wid * min_wid + bk * min_wid * min_wid / 2 + j * min_wid / 2 + i)] = 1;
        r_idx[3 * b * G * min_wid * min_wid / 2] = tmp;
    }
}
// conv.go:374-414: sum over (rot, mask) of Rotate(ct * mask, rot), twice (masks at scale sqrt(q_level)), one rescale
static DCt ext_double_ctxt(Boot *B, const DCt &ct, const IdxMap &m_idx, const IdxMap &r_idx, const std::string &key) {
    const double sq = sqrt((double)B->Q[(size_t)ct.level]);
    auto stage = [&](const DCt &in, const IdxMap &idx, const char *which) {
        DCt acc; bool have = false;
        for (auto &e : idx) {
            DCt t = B->rotate(B->mul_plain(in, B->encode_mask(key + which + std::to_string(e.first), e.second, in.level, sq)), e.first);
            acc = have ? B->add(acc, t) : t; have = true;
        }
        return acc;
    };
    return B->rescale(stage(stage(ct, m_idx, "/m"), r_idx, "/r"));
}
// conv.go:435-480
static DCt evalReLU(Boot *B, const DCt &ct_in, double alpha) {
    const double a = (alpha + 1) / 2.0, b = (1 - alpha) / 2.0, sc = 1073741824.0;
    const std::vector<double> c1 = {0.0, 10.8541842577442, 0.0, -62.2833925211098, 0.0, 114.369227820443, 0.0, -62.8023496973074};
    const std::vector<double> c2 = {0.0, 4.13976170985111, 0.0, -5.84997640211679, 0.0, 2.94376255659280, 0.0, -0.454530437460152};
    std::vector<double> c3 = {0.0, 3.29956739043733, 0.0, -7.84227260291355, 0.0, 12.8907764115564, 0.0, -12.4917112584486, 0.0, 6.94167991428074, 0.0, -2.04298067399942, 0.0, 0.246407138926031};
    for (auto &v : c3) v *= b;
    printf("Eval: ");
    DCt s = B->eval_poly(ct_in, c1, sc, false);
    s = B->eval_poly(s, c2, sc, false);
    s = B->eval_poly(s, c3, sc, false);
    s = B->add_const(s, a);
    return B->mul_relin(s, Boot::drop_to(ct_in, s.level));            // Mul + Relinearize, no rescale (conv.go:475-477)
}
// conv.go:417-431
static DCt keep_ctxt(Boot *B, const DCt &ct, const std::vector<int> &idx, const std::string &key) {
    DPt pt = B->encode_mask(key, idx, ct.level, (double)B->Q[(size_t)ct.level]);
    return B->rescale(B->mul_plain(ct, pt));
}

// ---------------------------------------------------------------- public surface (hconv_host.hpp)
Boot *newBoot(const std::vector<int64_t> &sk, const Seed256 &seed, int device, const std::vector<int> &log_sparse_sets, int image_batch) {
    if (image_batch < 1 || image_batch > 8) panic("image batch must be 1..8");
    Boot *b = new Boot(); b->nb_max = image_batch;
    // a full-slot bootstrapper (kind "Conv": two halves per image) gets blocks for twice the batch when that fits the library's 8 images per launch: merge2
    if (std::find(log_sparse_sets.begin(), log_sparse_sets.end(), 0) != log_sparse_sets.end() && log_sparse_sets.size() == 1 && 2 * image_batch <= 8 && !(getenv("HCONV_NO_MERGE") && atoi(getenv("HCONV_NO_MERGE")))) { b->nb_max = 2 * image_batch; b->merge_parts = true; }
    b->build(sk, seed, device);
    for (int ls : log_sparse_sets) b->set(ls);
    return b;
}
// main.go:163-215: the rotations of the stride layers' ext_double_ctxt belong to the evaluator's rotation keys
void bootPrepareCompress(Boot *B, int in_wid, int kp_wid, int log_sparse) {
    for (int ul = 0; ul < (log_sparse ? 1 : 2); ul++) {
        IdxMap m_idx, r_idx; gen_comprs_sparse(N / 2, in_wid, kp_wid, log_sparse, ul, m_idx, r_idx);
        for (auto *m : {&m_idx, &r_idx}) for (auto &e : *m) { const int k = ((e.first % B->n) + B->n) % B->n; if (k) B->key(B->gal_rot(k), LV_RELU_TOP - 10); }
    }
}
void freeBoot(Boot *b) {
    if (!b) return;
    b->sets.clear(); b->pt_cache.clear(); b->mono_i.reset();      // every block returns to the pool
    for (uint64_t *blk : b->pool) hc_free(b->hc, blk);
    b->pool.clear();
    for (uint64_t *blk : b->pool_qp) hc_free(b->hc, blk);
    b->pool_qp.clear();
    for (uint64_t *blk : b->pool1) hc_free(b->hc, blk);
    b->pool1.clear();
    if (b->d_sk) hc_free(b->hc, b->d_sk);
    hc_ctx_destroy(b->hc); delete b;      // the switching keys are owned by the context
}

// HCONV_PROFILE=1: per-kernel HIP-event totals of one layer's tail (hc_set_option("profile")) on stderr; event records between
// launches perturb the stream, so the printed wall times of such a run are not the ones to quote
static void profile_dump(Boot *B, const char *label) {
    hc_ctx *hc = B->hc; char names[8192]; HCR(hc_sync(hc));
    if (hc_profile_names(hc, names, sizeof names)) return;
    std::vector<std::pair<double, std::string>> rows; double tot = 0; long nl = 0;
    for (char *tok = strtok(names, ","); tok; tok = strtok(nullptr, ",")) { double ms = 0; long n = 0; hc_profile_get(hc, tok, &ms, &n); char b[160]; snprintf(b, sizeof b, "%-28s %6ld launches %8.3f ms %7.1f us/launch", tok, n, ms, n ? 1e3 * ms / (double)n : 0.0); rows.push_back({ms, b}); tot += ms; nl += n; }
    std::sort(rows.rbegin(), rows.rend());
    fprintf(stderr, "[profile] %s: %ld launches, %.3f ms of kernels\n", label, nl, tot);
    for (auto &r : rows) fprintf(stderr, "[profile]   %s\n", r.second.c_str());
    hc_profile_get(hc, nullptr, nullptr, nullptr);
}

// test mode (HCONV_CHAIN_REPLAY / HCONV_CHAIN_REPLAY_BL): SHA-256 of each polynomial's rows 0..level, as gotrace's emit_ct; one line per image of the
// batch ("replay digest" for the first, "replay digest[z]" for image z > 0; `first_image` = the number of the batch's first image)
static void replay_digest_line(Boot *B, const char *what, const DCt &c, int first_image = 0) {
    hc_ctx *hc = B->hc;
    for (int z = 0; z < B->nb; z++) {
        const int img = first_image + z;
        std::string line = std::string("replay digest") + (img ? "[" + std::to_string(img) + "] " : " ") + what + " level " + std::to_string(c.level);
        char sc[40]; snprintf(sc, sizeof sc, " scale %.17g", c.scale); line += sc;
        for (int d = 0; d <= c.deg; d++) { std::vector<uint64_t> rows((size_t)(c.level + 1) * N); HCR(hc_download(hc, rows.data(), c.p[d].get() + (size_t)z * B->poly_stride(), rows.size() * 8)); B->unpack_rows(rows, c.level + 1); Sha256 h; h.update(rows.data(), rows.size() * 8); line += " " + h.hex(); }
        printf("%s\n", line.c_str());
    }
}
// eval.go:437-565: everything after the convolution(s). ct_conv = the level-0 convolution result at out_scale
// 2^(round(log2 Q0) - (pow+8)). kind "Conv" (log_sparse 0, two ciphertexts through sine/ReLU, keep_ctxt masks of gen_keep_vec),
// "Conv_sparse" (one packed ciphertext, gen_keep_vec_sparse), "StrConv_sparse" (one packed ciphertext, ext_double_ctxt with
// gen_comprs_sparse: kp_wid is then the NEXT block's raw width).
// The images of a batch (ct_conv_dev.size() <= the bootstrapper's HCONV_IMAGE_BATCH) go through the tail as ONE set of launches; results in the same order.
std::vector<BootCiphertext> evalConv_BNRelu_tail_batch(Boot *B, const std::string &kind, int log_sparse, const std::vector<const uint64_t *> &ct_conv_dev, double ct_scale, double alpha, double pow_, int in_wid, int kp_wid) {
    hc_ctx *hc = B->hc;
    const bool stride = kind == "StrConv_sparse", sparse = kind == "Conv_sparse" || stride;
    if (!sparse && kind != "Conv") panic("No kind!");
    if (!sparse && log_sparse != 0) panic("No cases for log_sparse");
    const int nimg = (int)ct_conv_dev.size();
    if (nimg < 1 || nimg > (B->merge_parts ? B->nb_max / 2 : B->nb_max)) panic("evalConv_BNRelu_tail: more images than the bootstrapper's image batch (HCONV_IMAGE_BATCH)");
    Boot::Batch batch_scope(B, nimg); B->alg_ct = B->alg_shared = 0;                                  // hc_set_batch(nimg) here, hc_set_batch(1) wherever this function is left
    DCt ct = B->new_ct(0, 1, ct_scale * pow(2.0, pow_));                                            // eval.go:437
    for (int z = 0; z < nimg; z++) for (int d = 0; d < 2; d++) HCR(hc_copy(hc, ct.p[d].get() + (size_t)z * B->poly_stride(), ct_conv_dev[(size_t)z] + (size_t)d * N, (size_t)N * 8));
    // ct_conv_dev belongs to ANOTHER context (the convolution's): the copies above are queued on this context's stream, and the caller frees
    // the sources as soon as this function returns. Wait for them (the stream holds nothing else at this point) so that the hand-over
    // does not depend on how the two contexts' streams happen to be scheduled (cached allocations recycle a freed block at once).
    HCR(hc_sync(hc));
    // test mode: image z of the batch is image first_image + z of the replay; image 0 carries the input gotrace -chain plants at the entry of
    // BootstrappConv_CtoS - SEED_OPIN(4000, 0, poly, limb 0) -, image i > 0 the same generator keyed seed + 0x1000003 i (keys are common)
    const int first_image = B->replay_seed && testOnlyEnv("HCONV_REPLAY_IMAGE0") ? atoi(testOnlyEnv("HCONV_REPLAY_IMAGE0")) : 0;
    auto replay_digest = [&](const char *what, const DCt &c) { replay_digest_line(B, what, c, first_image); };
    if (B->replay_seed) {
        if (sparse) panic("HCONV_CHAIN_REPLAY covers the full-slot chain only");
        std::vector<uint64_t> row((size_t)N);
        for (int z = 0; z < nimg; z++) for (int k = 0; k < 2; k++) {
            const uint64_t sd = B->replay_seed + 0x1000003ull * (uint64_t)(first_image + z) + ((6ull << 32) | (uint64_t)((((4000 * 2 + 0) * 4 + k) * 64) + 0));
            for (int j = 0; j < N; j++) row[(size_t)j] = Boot::splitmix_at(sd, (uint64_t)j) % B->Q[0];
            HCR(hc_upload(hc, ct.p[k].get() + (size_t)z * B->poly_stride(), row.data(), (size_t)N * 8));
        }
    }
    const bool prof = getenv("HCONV_PROFILE") && atoi(getenv("HCONV_PROFILE"));
    if (prof) { HCR(hc_set_option(hc, "profile", 1)); hc_profile_get(hc, nullptr, nullptr, nullptr); }
    printf("Bootstrapping... Ours (until CtoS):\n");
    auto start = now();
    // test mode, with HCONV_CHAIN_REPLAY: HCONV_REPLAY_LOG_SPARSE=ls runs BootstrappConv_CtoS of the SPARSE-slot bootstrapper on the planted input, as
    // `gotrace -chain -logslots 15-ls` made the reference binary do inside this very `convReLU` run (tests/golden/ref_trace_chain_sparse_ls13.json);
    // the binary's run ends there (it panics on the nil second result), so does this one
    const char *rls = B->replay_seed ? testOnlyEnv("HCONV_REPLAY_LOG_SPARSE") : nullptr;
    const int ls_run = rls ? atoi(rls) : log_sparse;
    DCt boots[2]; const int iter = B->ctos(ct, ls_run, boots);                                         // eval.go:450-461
    HCR(hc_sync(hc));
    printf("Done in %s \n", dur(start).c_str());
    if (prof) profile_dump(B, "sine");
    if (B->replay_seed) {
        if (B->parts_merged) { DCt both = boots[0]; B->split2(both, boots); B->parts_merged = false; }      // test mode prints the halves: continue unmerged
        for (int ul = 0; ul < iter; ul++) replay_digest(ul ? "ctos1" : "ctos0", boots[ul]);
    }
    if (rls) { printf("replay of the sparse-slot BootstrappConv_CtoS done (log_sparse %d)\n", ls_run); fflush(stdout); exit(0); }
    start = now();
    for (int ul = 0; ul < (B->parts_merged ? 1 : iter); ul++) {
        if (B->parts_merged) printf("Eval: ");                                                        // the reference's loop (eval.go:462-477) prints once per half: same line shape
        DCt r = evalReLU(B, boots[ul], alpha);                                                        // eval.go:473
        boots[ul] = B->mul_const_int(r, pow(2.0, pow_));                                              // MulByPow2 (eval.go:474)
    }
    if (B->parts_merged) { DCt both = boots[0]; B->split2(both, boots); B->parts_merged = false; }    // the halves part again: their masks differ
    HCR(hc_sync(hc));
    printf("ReLU Done in %s \n", dur(start).c_str());
    if (prof) profile_dump(B, "ReLU");
    start = now();
    DCt keep[2];
    const std::string mk = std::to_string(in_wid) + "/" + std::to_string(kp_wid) + "/" + std::to_string(log_sparse);
    if (stride) for (int ul = 0; ul < iter; ul++) {                                                                                                                            // eval.go:500-506: m_idx / m_idx_l
        IdxMap m_idx, r_idx; gen_comprs_sparse(N / 2, in_wid, kp_wid, log_sparse, ul, m_idx, r_idx); keep[ul] = ext_double_ctxt(B, boots[ul], m_idx, r_idx, "comprs/" + mk + "/" + std::to_string(ul)); }
    else if (sparse && log_sparse) keep[0] = keep_ctxt(B, boots[0], gen_keep_vec_sparse(N / 2, in_wid, kp_wid, log_sparse), "keep/" + mk);                                         // eval.go:534
    // "Conv_sparse" on full packing (wide_case 3, block 1) keeps with gen_keep_vec per half, like "Conv" (main.go:155-156)
    else for (int ul = 0; ul < 2; ul++) keep[ul] = keep_ctxt(B, boots[ul], gen_keep_vec(N / 2, in_wid, kp_wid, ul), "keep/" + mk + "/" + std::to_string(ul));
    DCt res = B->stoc(keep[0], iter == 2 ? &keep[1] : nullptr, log_sparse);                              // eval.go:550-561 ; Rescale (564) is a no-op here
    HCR(hc_sync(hc));
    printf("Boot (StoC) Done in %s \n", dur(start).c_str());
    if (B->replay_seed) replay_digest("final", res);
    if (prof) { profile_dump(B, ("mask + SlotsToCoeffs; the layer was " + kind + " log_sparse " + std::to_string(log_sparse)).c_str()); HCR(hc_set_option(hc, "profile", 0)); }
    if (getenv("HCONV_ALG_BYTES")) printf("algorithmic traffic of the layer's tail: %.6g GB per ciphertext + %.6g GB shared by the %d image%s of the launch set\n", B->alg_ct * N * 8 / 1e9, B->alg_shared * N * 8 / 1e9, nimg, nimg > 1 ? "s" : "");
    B->alg_ct = B->alg_shared = 0;
    std::vector<BootCiphertext> outs((size_t)nimg);
    for (int z = 0; z < nimg; z++) {
        BootCiphertext &out = outs[(size_t)z]; out.level = res.level; out.Scale = res.scale;
        { void *v = nullptr; HCR(hc_malloc(hc, (size_t)2 * (res.level + 1) * N * 8, &v)); out.d = (uint64_t *)v; }
        for (int d = 0; d < 2; d++) HCR(hc_copy(hc, out.d + (size_t)d * (res.level + 1) * N, res.p[d].get() + (size_t)z * B->poly_stride(), (size_t)(res.level + 1) * N * 8));
    }
    HCR(hc_sync(hc));
    return outs;
}
BootCiphertext evalConv_BNRelu_tail(Boot *B, const std::string &kind, int log_sparse, const uint64_t *ct_conv_dev, double ct_scale, double alpha, double pow_, int in_wid, int kp_wid) {
    return evalConv_BNRelu_tail_batch(B, kind, log_sparse, {ct_conv_dev}, ct_scale, alpha, pow_, in_wid, kp_wid)[0];
}
// ---------------------------------------------------------------- baseline: Bootstrapp + ReLU (test_BL.go:113-168)
Boot *newBootBL(const std::vector<int64_t> &sk, const Seed256 &seed, int device) {
    Boot *b = new Boot(); b->build(sk, seed, device, 7); b->set(0);
    return b;
}
void blBootReLU(Boot *B, const uint64_t *ct_res0, const uint64_t *ct_res1, double scale, double alpha, double pow_, uint64_t *out0, uint64_t *out1, double *out_scale) {
    hc_ctx *hc = B->hc;
    if (B->chain != 7) panic("blBootReLU needs the baseline's bootstrapper (newBootBL)");
    auto img_eval = now();
    DCt c[2];
    for (int pos = 0; pos < 2; pos++) {
        DCt t = B->new_ct(1, 1, scale); const uint64_t *src = pos ? ct_res1 : ct_res0;
        for (int d = 0; d < 2; d++) HCR(hc_copy(hc, t.p[d].get(), src + (size_t)d * 2 * N, (size_t)2 * N * 8));
        c[pos] = B->add(B->conjugate(t), t);                                                             // test_BL.go:116
        if (pos == 1) c[pos] = B->mul_by_i(c[pos]);                                                      // MultByiNew (118)
    }
    DCt ct = B->add(c[0], c[1]);                                                                         // test_BL.go:122
    HCR(hc_sync(hc));
    const double img_part = std::chrono::duration<double>(now() - img_eval).count();
    ct.scale = ct.scale * exp2(pow_ + 2);                                                                // test_BL.go:128
    printf("\n ========= Bootstrapping... (original) ========= \n");
    auto start_boot = now();
    // ckks.(*Bootstrapper).Bootstrapp, stock, op for op as the reference binary runs it (gotrace -flow-bl: tests/golden/ref_flow_bl_5_1.json; on planted data: ref_trace_chain_bl_5_1.json):
    // SetScale to 2^round(log2(Q0 / MessageRatio)) (MessageRatio 256) down to level 0, modUp, CoeffsToSlots, the sine on both halves, SlotsToCoeffs on levels 15, 15, 14
    const bool dbg = getenv("HCONV_DEBUG_BOOT") && *getenv("HCONV_DEBUG_BOOT"); std::vector<cplx> z_in;
    if (dbg) z_in = B->debug_slots(ct);
    if (B->replay_seed) {            // the input gotrace -flow-bl -chain plants at the entry of Bootstrapp: SEED_OPIN(4001, 0, poly, limb), both limbs of the level-1 ciphertext
        std::vector<uint64_t> rows((size_t)2 * N);
        for (int k = 0; k < 2; k++) {
            for (int l = 0; l < 2; l++) { const uint64_t sd = B->replay_seed + ((6ull << 32) | (uint64_t)((((4001 * 2 + 0) * 4 + k) * 64) + l));
                for (int j = 0; j < N; j++) rows[(size_t)l * N + (size_t)j] = Boot::splitmix_at(sd, (uint64_t)j) % B->Q[(size_t)l]; }
            HCR(hc_upload(hc, ct.p[k].get(), rows.data(), rows.size() * 8));
        }
    }
    DCt halves[2]; if (B->ctos(ct, 0, halves) != 2) panic("Bootstrapp: full-slot CoeffsToSlots returns two ciphertexts");
    if (B->replay_seed) for (int h = 0; h < 2; h++) replay_digest_line(B, h ? "sine1" : "sine0", halves[h]);
    DCt ct_boot = B->stoc(halves[0], &halves[1], 0);
    if (B->replay_seed) { replay_digest_line(B, "bootstrapp", ct_boot); printf("replay of the baseline's Bootstrapp done\n"); fflush(stdout); exit(0); }
    if (dbg) Boot::debug_compare("Bootstrapp", z_in, B->debug_slots(ct_boot));
    HCR(hc_sync(hc));
    printf("Boot Done in %s \n", dur(start_boot).c_str());
    img_eval = now();
    // test_BL.go:146-153: an all-ones plaintext at scale 2^30 q14 q13 / ct_boot.Scale, Mul, Rescale(params.Scale): level 14, scale ~2^120 -> level 12, scale 2^30
    { const int L = ct_boot.level; if (L != 14) panic("Bootstrapp ended at an unexpected level");
      std::vector<cplx> ones((size_t)N / 2, cplx(1.0, 0)); DPt pl_scale = B->encode(ones, L, 1073741824.0 * (double)B->Q[14] * (double)B->Q[13] / ct_boot.scale);
      ct_boot = B->lt_rescale(B->mul_plain(ct_boot, pl_scale), 1073741824.0);
      if (ct_boot.level != 12) panic("the baseline's rescale after Bootstrapp ended at an unexpected level"); }
    if (dbg) Boot::debug_compare("Bootstrapp + scale plaintext", z_in, B->debug_slots(ct_boot));
    DCt ct_iboot = B->conjugate(ct_boot);                                                                // test_BL.go:155
    DCt res[2] = {B->add(ct_boot, ct_iboot), B->mul_by_i(B->sub(ct_iboot, ct_boot))};                    // DivByi(a - b) = i (b - a)
    HCR(hc_sync(hc));
    { double ns = (img_part + std::chrono::duration<double>(now() - img_eval).count()) * 1e9; char b[64];
      if (ns < 1e6) snprintf(b, sizeof b, "%.6gµs", ns / 1e3); else if (ns < 1e9) snprintf(b, sizeof b, "%.9gms", ns / 1e6); else snprintf(b, sizeof b, "%.9gs", ns / 1e9);
      printf("Imaginary packing and unpacking done in %s \n", b); }
    auto start = now();
    for (int pos = 0; pos < 2; pos++) {                                                                  // test_BL.go:160-167
        DCt r = evalReLU(B, res[pos], alpha);
        r = B->mul_const_int(r, pow(2.0, pow_));                                                         // MulByPow2
        r = B->set_scale(r, 1073741824.0);
        if (r.level != 1) panic("baseline ReLU ended at an unexpected level");
        uint64_t *dst = pos ? out1 : out0;
        for (int d = 0; d < 2; d++) HCR(hc_copy(hc, dst + (size_t)d * 2 * N, r.p[d].get(), (size_t)2 * N * 8));
        *out_scale = r.scale;
    }
    HCR(hc_sync(hc));
    printf("Relu Done in %s \n", dur(start).c_str());
}

// Decrypt at level 1 + DecodeCoeffs: CRT over Q0*Q1, centre, / scale
std::vector<double> bootDecryptDecodeCoeffs(Boot *B, const BootCiphertext &ct) {
    hc_ctx *hc = B->hc;
    if (ct.level != 1) panic("bootDecryptDecodeCoeffs expects the level-1 result of the chain");
    void *v = nullptr; HCR(hc_malloc(hc, (size_t)2 * N * 8, &v)); uint64_t *t = (uint64_t *)v;
    HCR(hc_lv_mul_plain(hc, 1, ct.d + (size_t)2 * N, B->d_sk, t)); HCR(hc_lv_add(hc, 1, ct.d, t, t)); HCR(hc_lv_intt(hc, 1, t, t));
    std::vector<uint64_t> m((size_t)2 * N); HCR(hc_download(hc, m.data(), t, m.size() * 8)); HCR(hc_free(hc, t));
    const uint64_t q0 = B->Q[0], q1 = B->Q[1]; uint64_t inv = 1; { uint64_t b = q0 % q1, e = q1 - 2; while (e) { if (e & 1) inv = mulmod(inv, b, q1); b = mulmod(b, b, q1); e >>= 1; } }
    const u128 QQ = (u128)q0 * q1; std::vector<double> cf((size_t)N);
    for (int j = 0; j < N; j++) {
        const uint64_t a0 = m[(size_t)j], a1 = m[(size_t)N + j], d = (a1 % q1 + q1 - a0 % q1) % q1;
        const u128 x = (u128)a0 + (u128)q0 * mulmod(d, inv, q1);
        cf[(size_t)j] = x > QQ / 2 ? -(double)(QQ - x) / ct.Scale : (double)x / ct.Scale;
    }
    return cf;
}
void freeBootCt(Boot *B, BootCiphertext &ct) { hc_ctx *hc = B->hc; if (ct.d) HCR(hc_free(hc, ct.d)); ct.d = nullptr; }
void bootStats(Boot *B, long *keys, long *keyswitches) { *keys = B->n_keys; *keyswitches = B->n_keyswitch; }

}  // namespace hconv
