// hconv_host.cpp — see hconv_host.hpp for the reference mapping. Everything that touches residues of a ciphertext
// on the conv path runs on the GPU through the C ABI; the host does float index shuffling, sampling and printing.
#include "hconv_host.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <random>
#include <sstream>

namespace hconv {

const std::vector<uint64_t> PARAMS6_Q = {
    0x80000000080001ull, 0x1ffffffea0001ull,                                                          // residual (levels 0-1)
    0x1000000000b00001ull, 0x1000000000ce0001ull,                                                     // StC
    0x3ffffe80001ull,                                                                                 // 42-bit
    0x3ffc0001ull, 0x40080001ull, 0x3fac0001ull, 0x40720001ull, 0x3f820001ull, 0x3f760001ull, 0x40980001ull,
    0x3f5a0001ull, 0x3f540001ull, 0x40b00001ull, 0x40c20001ull,                                       // 11 x ~30-bit
    0x80000000440001ull, 0x7fffffffba0001ull, 0x80000000500001ull, 0x7fffffffaa0001ull, 0x800000005e0001ull,
    0x7fffffff7e0001ull, 0x7fffffff380001ull, 0x80000000ca0001ull,                                    // sine
    0x200000000e0001ull, 0x20000000140001ull, 0x20000000280001ull, 0x1fffffffd80001ull};              // CtS
// ckks.DefaultBootstrapParams[7] (the baseline's set, main.go:54): residual group = levels 0-13, StC 14-15, sine, CtS
const std::vector<uint64_t> PARAMS7_Q = {
    0x80000000080001ull, 0x10000000006e0001ull,
    0x3ffc0001ull, 0x40080001ull, 0x3fac0001ull, 0x40720001ull, 0x3f820001ull, 0x3f760001ull, 0x40980001ull,
    0x3f5a0001ull, 0x3f540001ull, 0x40b00001ull, 0x40c20001ull,                                       // 11 x ~30-bit (levels 2-12)
    0xffffffffffc0001ull,                                                                             // level 13
    0x1000000000b00001ull, 0x1000000000ce0001ull,                                                     // StC
    0x80000000440001ull, 0x7fffffffba0001ull, 0x80000000500001ull, 0x7fffffffaa0001ull, 0x800000005e0001ull,
    0x7fffffff7e0001ull, 0x7fffffff380001ull, 0x80000000ca0001ull,                                    // sine
    0x200000000e0001ull, 0x20000000140001ull, 0x20000000280001ull, 0x1fffffffd80001ull};              // CtS
const std::vector<uint64_t> PARAMS6_P = {0x1fffffffffe00001ull, 0x1fffffffffc80001ull, 0x1fffffffffb40001ull,
                                         0x1fffffffff500001ull, 0x1fffffffff420001ull};

void panic(const std::string &msg) {
    fprintf(stderr, "panic: %s\n", msg.c_str());
    exit(2);
}
bool &testMode() { static bool on = false; return on; }
const char *testOnlyEnv(const char *name) {
    const char *v = getenv(name);
    if (!v || !*v) return nullptr;
    if (!testMode()) panic(std::string(name) + " is set but the CLI was not started with --test-mode: refusing to run with test-only key / input overrides");
    return v;
}
#define HC(c, call) do { int rc_ = (call); if (rc_) panic(std::string(#call) + ": " + hc_last_error(c)); } while (0)

static const uint64_t MODQ[3] = {0x80000000080001ull, 0x1ffffffea0001ull, PACK_P};   // Q0, Q1, P (ABI indices 0,1,2)

static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t r = a + b; return r >= q ? r - q : r; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static inline uint64_t to_mont(uint64_t a, uint64_t q) { return (uint64_t)((((u128)a) << 64) % q); }

// Go's time.Duration.String() shape ("57.154502ms", "3.40439908s")
static std::string dur(std::chrono::steady_clock::time_point t0) {
    double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    char b[64];
    if (ns < 1e3) snprintf(b, sizeof b, "%.0fns", ns);
    else if (ns < 1e6) snprintf(b, sizeof b, "%.6gµs", ns / 1e3);
    else if (ns < 1e9) snprintf(b, sizeof b, "%.9gms", ns / 1e6);
    else snprintf(b, sizeof b, "%.9gs", ns / 1e9);
    return b;
}
static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }

// ---------------------------------------------------------------- device helpers
static uint64_t *dev_rows(Context *c, size_t rows) { void *p = nullptr; HC(c->hc, hc_malloc(c->hc, rows * N * sizeof(uint64_t), &p)); return (uint64_t *)p; }
static uint64_t *dev_upload(Context *c, const std::vector<uint64_t> &h) { uint64_t *d = dev_rows(c, h.size() / N); HC(c->hc, hc_upload(c->hc, d, h.data(), h.size() * 8)); return d; }
static std::vector<uint64_t> dev_download(Context *c, const uint64_t *d, size_t rows) { std::vector<uint64_t> h(rows * N); HC(c->hc, hc_download(c->hc, h.data(), d, h.size() * 8)); return h; }
// NTT of host rows (all rows mod the same modulus) on the GPU
static std::vector<uint64_t> gpu_ntt(Context *c, int mod, const std::vector<uint64_t> &rows, bool inverse = false) {
    uint64_t *d = dev_upload(c, rows); int cnt = (int)(rows.size() / N);
    HC(c->hc, inverse ? hc_intt(c->hc, mod, d, d, cnt) : hc_ntt(c->hc, mod, d, d, cnt));
    std::vector<uint64_t> out = dev_download(c, d, (size_t)cnt);
    HC(c->hc, hc_free(c->hc, d));
    return out;
}

// ---------------------------------------------------------------- test mode: the oracle harness' deterministic generators (HCONV_RESNET_REPLAY)
uint64_t resnetReplaySeed() { static const uint64_t v = testOnlyEnv("HCONV_RESNET_REPLAY") ? strtoull(testOnlyEnv("HCONV_RESNET_REPLAY"), nullptr, 0) : 0; return v; }
namespace replay {
uint64_t sm64(uint64_t seed, uint64_t i) { uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
void fill_seeded(uint64_t seed, uint64_t q, uint64_t *out) { for (int j = 0; j < N; j++) out[j] = sm64(seed, (uint64_t)j) % q; }
void gauss(uint64_t seed, std::vector<int64_t> &e) {
    e.resize(N);
    for (int j = 0; j < N; j++) for (uint64_t k = 0;; k++) {
        const double u1 = ((double)(sm64(seed, (uint64_t)j * 64 + 2 * k) >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(sm64(seed, (uint64_t)j * 64 + 2 * k + 1) >> 11) / 9007199254740992.0;
        const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * 3.2;
        if (fabs(g) <= 19.2) { e[(size_t)j] = (int64_t)llround(g); break; }
    }
}
std::vector<int64_t> gen_sk(uint64_t seed, int h) {
    std::vector<int64_t> sk((size_t)N, 0); uint64_t ctr = 0; int placed = 0;
    while (placed < h) { const uint64_t r = sm64(seed, ctr++); const int pos = (int)(r % (uint64_t)N); if (!sk[(size_t)pos]) { sk[(size_t)pos] = (r >> 40) & 1 ? 1 : -1; placed++; } }
    return sk;
}
}  // namespace replay

// ---------------------------------------------------------------- sampling (harness only)
static ChaChaRng &rng(Context *c) { return c->g; }      // one generator per context (image threads own their contexts)
static std::vector<uint64_t> uniform_row(Context *c, uint64_t q) {
    std::vector<uint64_t> r(N); std::uniform_int_distribution<uint64_t> d(0, q - 1); auto &g = rng(c);
    for (auto &x : r) x = d(g);
    return r;
}
static std::vector<int64_t> gaussian(Context *c) {   // sigma = 3.2 (rlwe.DefaultSigma, main.go:421), bound 6 sigma
    std::vector<int64_t> e(N); std::normal_distribution<double> d(0.0, 3.2); auto &g = rng(c);
    for (auto &x : e) { double v; do { v = d(g); } while (fabs(v) > 19.2); x = (int64_t)llround(v); }
    return e;
}
static std::vector<uint64_t> signed_row(const std::vector<int64_t> &v, uint64_t q) {
    std::vector<uint64_t> r(N);
    for (int j = 0; j < N; j++) r[j] = v[j] >= 0 ? (uint64_t)v[j] % q : q - ((uint64_t)(-v[j]) % q);
    return r;
}

// rlwe.GenRotationKeys for one Galois element, restricted to what level-0 key switching reads (digit 0; limbs Q0, P):
// b = -a*sigma_{g^-1}(s) + e + P*s, stored NTT + Montgomery (SURVEY.md 8(a)-R last rows)
static void gen_and_load_galois_key(Context *c, uint64_t galEl) {
    const uint64_t twoN = 2ull * N; uint64_t ginv = 1, b = galEl % twoN;
    for (uint64_t e = twoN - 1; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN;
    std::vector<int64_t> sko(N, 0);
    for (int i = 0; i < N; i++) { uint64_t t = ((uint64_t)i * ginv) % twoN; if (t < (uint64_t)N) sko[t] = c->sk[i]; else sko[t - N] = -c->sk[i]; }
    const uint64_t rs = resnetReplaySeed(); int jj = 0; while ((1ull << jj) + 1 < galEl) jj++;       // replay: or_gen_galois_key_l0(sk, 2^j + 1, seed 100 + j) of the oracle network
    std::vector<int64_t> e; if (rs) replay::gauss((100 + (uint64_t)jj) ^ 0xE44E44ull, e); else e = gaussian(c);
    std::vector<uint64_t> rows[4];   // b_q, a_q, b_p, a_p
    const int mods[2] = {0, 2};
    for (int w = 0; w < 2; w++) {
        const int mod = mods[w]; const uint64_t q = MODQ[mod];
        std::vector<uint64_t> a; if (rs) { a.resize(N); replay::fill_seeded(100 + (uint64_t)jj + 0x1000 + (uint64_t)w, q, a.data()); } else a = uniform_row(c, q);
        std::vector<uint64_t> both = signed_row(sko, q), en = signed_row(e, q);
        both.insert(both.end(), en.begin(), en.end());
        both = gpu_ntt(c, mod, both);
        const uint64_t *so = both.data(), *ent = both.data() + N; const std::vector<uint64_t> &si = c->sk_ntt[mod];
        const uint64_t pmod = w == 0 ? PACK_P % q : 0;
        std::vector<uint64_t> bb(N);
        for (int j = 0; j < N; j++) {
            uint64_t v = submod(ent[j], mulmod(a[j], so[j], q), q);
            v = addmod(v, mulmod(pmod, si[j], q), q);
            bb[j] = to_mont(v, q); a[j] = to_mont(a[j], q);
        }
        rows[2 * w] = bb; rows[2 * w + 1] = a;
    }
    HC(c->hc, hc_evk_load(c->hc, galEl, rows[0].data(), rows[1].data(), rows[2].data(), rows[3].data()));
    for (size_t g = 1; g < c->shards.size(); g++) HC(c->shards[g], hc_evk_load(c->shards[g], galEl, rows[0].data(), rows[1].data(), rows[2].data(), rows[3].data()));
    if (galEl - 1 < 32) {       // 2^j+1 with j < 5 leaves the 4096-coefficient tile of the fused kernels: RotateGal takes the general key switch
        std::vector<uint64_t> g; for (int r : {0, 2, 1, 3}) g.insert(g.end(), rows[r].begin(), rows[r].end());     // [digit 0][b | a][Q0, P][N]
        HC(c->hc, hc_swk_load(c->hc, galEl, 0, g.data()));
    }
}

// ---------------------------------------------------------------- newContext (main.go:44-462, kind "Conv")
void applyEnvOptions(hc_ctx *hc, int pack32_default) {
    auto opt = [&](const char *env, const char *name, long dflt) {
        const char *v = getenv(env);
        const long val = (v && *v) ? atol(v) : dflt;
        if (val < 0) return;
        if (hc_set_option(hc, name, val)) panic(std::string("hc_set_option(") + name + "): " + hc_last_error(hc));
    };
    opt("HCONV_ASYNC_ALLOC", "async_alloc", -1);      // first: the allocation mode can only change while the context owns nothing but its tables
    opt("HCONV_SMALL32", "small32", -1);
    opt("HCONV_ROT_FUSE", "rot_fuse", -1);
    opt("HCONV_SMALL_MM_WGS", "small_mm_wgs", -1);    // A/B: 16-row workgroups per pass up to which the batched inverse transforms run on quarter tiles (0: never)
    opt("HCONV_PACK32", "pack32", pack32_default);
}

Context *newContext(int logN, int ker_wid, const std::vector<int> &in_wids, const std::vector<int> &kp_wids, bool boot, const std::string &kind) {
    (void)ker_wid;
    if (kind != "Conv" && kind != "Resnet_crop_sparse") panic("Wrong kinds!");       // main.go:404 (the kinds the built command lines use)
    Context *c = new Context();
    double logqp = 0; for (uint64_t q : PARAMS6_Q) logqp += log2((double)q); for (uint64_t p : PARAMS6_P) logqp += log2((double)p);
    printf("CKKS parameters: logN = %d, logSlots = %d, h = %d, logQP = %d, levels = %d, scale= 2^%f, sigma = %f \n",
           LOGN, LOGN - 1, 192, (int)llround(logqp), (int)PARAMS6_Q.size(), log2(c->scale), 3.2);       // main.go:85-86
    if ((1 << logN) != N) { printf("Set Boot logN to %d\n", logN); panic("Boot N != N"); }           // main.go:87-90
    c->seed = seedFromEnvironment(); c->g.reseed(c->seed, 0x436f6e76);      // 256 bits from getrandom(2) unless HCONV_SEED asks for deterministic keys
    int dev = getenv("HCONV_DEVICE") ? atoi(getenv("HCONV_DEVICE")) : 0;
    uint64_t q[2] = {MODQ[0], MODQ[1]}, p[1] = {PACK_P};
    if (hc_ctx_create(&c->hc, LOGN, q, 2, p, 1, dev)) panic(std::string("hc_ctx_create: ") + hc_last_error(nullptr));
    applyEnvOptions(c->hc);
    c->shards.push_back(c->hc);
    if (const char *ng = getenv("HCONV_GPUS")) {       // one convolution over G devices (contexts share devices when the box has fewer)
        const int G = atoi(ng); int ndev = 1; hc_device_count(&ndev);
        if (G < 1 || G > 16 || (G & (G - 1))) panic("HCONV_GPUS must be a power of two in 1..16");
        for (int g = 1; g < G; g++) { hc_ctx *h = nullptr; if (hc_ctx_create(&h, LOGN, q, 2, p, 1, (dev + g) % ndev)) panic(std::string("hc_ctx_create: ") + hc_last_error(nullptr)); applyEnvOptions(h); c->shards.push_back(h); }
        if (G > 1) printf("Sharding every convolution over %d device contexts (%d device%s visible)\n", G, ndev, ndev == 1 ? "" : "s");
    }
    // kgen.GenKeyPairSparse(h = 192) (main.go:410)
    c->sk.assign(N, 0);
    if (resnetReplaySeed()) c->sk = replay::gen_sk(resnetReplaySeed(), 192);       // the oracle network's key: Ckks(seed).sk = or_gen_sk(seed, 192)
    else { auto &g = rng(c); int placed = 0; while (placed < 192) { uint64_t r = g(); int pos = (int)(r % N); if (!c->sk[pos]) { c->sk[pos] = (r >> 40) & 1 ? 1 : -1; placed++; } } }
    for (int m = 0; m < 3; m++) c->sk_ntt[m] = gpu_ntt(c, m, signed_row(c->sk, MODQ[m]));
    printf("Num Rotations:  %d\n", c->num_rotations);                                                  // main.go:412
    // gen_idxNlogs (conv.go:241-261): idx[i] = NTT(X^(2^i)) on the device; Galois keys for 2^(i+1)+1, i < logN
    for (hc_ctx *h : c->shards) HC(h, hc_idx_load(h, nullptr));
    for (int i = 0; i < LOGN; i++) gen_and_load_galois_key(c, (1ull << (i + 1)) + 1);
    if (boot) {                                                                                        // main.go:464-507
        printf("Generating bootstrapping keys...\n");
        auto start = now();
        // DFT matrices and every switching key, same secret key. "Conv": one full-slot bootstrapper; the resnet kind: the
        // four sparse ones its layers use (btp2..btp5 of main.go:480-500; log_sparse 1..4)
        c->btp = kind == "Conv" ? newBoot(c->sk, c->seed, dev, {0}, imageBatch()) : newBoot(c->sk, c->seed, dev, std::vector<int>{2, 1, 3, 4}, imageBatch());
        if (kind != "Conv") { bootPrepareCompress(c->btp, in_wids[0], kp_wids[1], 1); bootPrepareCompress(c->btp, in_wids[1], kp_wids[2], 2); }   // main.go:163-215
        printf("Done in %s \n", dur(start).c_str());
    }
    return c;
}
void freeContext(Context *c) { if (!c) return; freeBoot(c->btp); for (size_t g = 1; g < c->shards.size(); g++) hc_ctx_destroy(c->shards[g]); hc_ctx_destroy(c->hc); delete c; }

// ---------------------------------------------------------------- text I/O and float layout
std::vector<double> readTxt(const std::string &name_file, int size) {      // main.go:971-990
    std::ifstream f(name_file);
    if (!f) panic("open " + name_file + ": no such file or directory");
    std::vector<double> input; std::string w;
    while (f >> w) input.push_back(strtod(w.c_str(), nullptr));
    if (size != 0 && (int)input.size() != size) panic("input size inconsistent!");
    return input;
}
std::vector<double> prep_Input(const std::vector<double> &input, int raw_in_wid, int in_wid, int Nn, int norm, bool trans, bool) {   // main.go:1007-1042
    if (trans) panic("transposed convolution is outside the conv CLI path");
    std::vector<double> out((size_t)Nn, 0.0); int batch = Nn / (in_wid * in_wid), k = 0;
    for (int i = 0; i < in_wid; i++) for (int j = 0; j < in_wid; j++) for (int b = 0; b < batch / norm; b++)
        if (i < raw_in_wid && j < raw_in_wid) out[(size_t)(i * in_wid * batch + j * batch + b * norm)] = input[(size_t)k++];
    return out;
}
std::vector<std::vector<double>> reshape_ker(const std::vector<double> &ker_in, int k_sz, int out_batch, bool trans) {   // conv.go:184-202
    if (trans) panic("transposed convolution is outside the conv CLI path");
    int in_batch = (int)ker_in.size() / (k_sz * out_batch);
    std::vector<std::vector<double>> ker_out((size_t)out_batch, std::vector<double>((size_t)(k_sz * in_batch)));
    for (int i = 0; i < out_batch; i++) for (int j = 0; j < in_batch; j++) for (int k = 0; k < k_sz; k++)
        ker_out[(size_t)i][(size_t)(j * k_sz + k)] = ker_in[(size_t)(i + j * out_batch + k * out_batch * in_batch)];
    return ker_out;
}
std::vector<double> encode_ker_final(const std::vector<std::vector<double>> &ker_in, int pos, int i, int in_wid, int in_batch, int ker_wid) {   // conv.go:206-237
    int vec_size = in_wid * in_wid * in_batch, k_sz = ker_wid * ker_wid, bias = pos * ker_wid * ker_wid * in_batch;
    std::vector<double> output((size_t)vec_size, 0.0);
    for (int j = 0; j < in_batch; j++) for (int k = 0; k < k_sz; k++)
        output[(size_t)((in_wid * (k / ker_wid) + k % ker_wid) * in_batch + j)] = ker_in[(size_t)i][(size_t)((in_batch - 1 - j) * k_sz + (k_sz - 1 - k) + bias)];
    int adj = (in_batch - 1) + in_batch * (in_wid + 1) * (ker_wid - 1) / 2;
    std::vector<double> tmp((size_t)adj);
    for (int t = 0; t < adj; t++) { tmp[(size_t)t] = output[(size_t)(vec_size - adj + t)]; output[(size_t)(vec_size - adj + t)] = -output[(size_t)t]; }
    for (int t = 0; t < vec_size - 2 * adj; t++) output[(size_t)t] = output[(size_t)(t + adj)];
    for (int t = 0; t < adj; t++) output[(size_t)(t + vec_size - 2 * adj)] = tmp[(size_t)t];
    return output;
}
std::vector<double> post_process(const std::vector<double> &in_cfs, int raw_in_wid, int in_wid) {     // main.go:1057-1070
    int batch = (int)in_cfs.size() / (in_wid * in_wid);
    std::vector<double> out((size_t)(raw_in_wid * raw_in_wid * batch));
    for (int i = 0; i < raw_in_wid; i++) for (int j = 0; j < raw_in_wid; j++) for (int b = 0; b < batch; b++)
        out[(size_t)(i * raw_in_wid * batch + batch * j + b)] = in_cfs[(size_t)(i * in_wid * batch + batch * j + b)];
    return out;
}
void set_Variables(int batch, int raw_in_wid, int in_wid, int ker_wid, const std::string &kind, int *kp_wid, int *out_batch, int *logN, bool *trans) {   // eval.go:13-54
    int Nn = batch * in_wid * in_wid; *logN = 0; while ((1 << *logN) < Nn) (*logN)++;
    int max_kp_wid = in_wid - ((ker_wid - 1) / 2);
    if (kind != "Conv") panic("Wrong kinds!");
    *trans = false; *kp_wid = raw_in_wid; *out_batch = batch;
    if (*kp_wid > max_kp_wid) { printf("max raw_in_wid:  %d\n", max_kp_wid); panic("too large raw_in_wid."); }
}
void printDebugCfsPlain(const std::vector<double> &valuesTest, const std::vector<double> &valuesWant) {   // main.go:694-717
    printf("ValuesTest:"); for (int i = 0; i < 10; i++) printf("%6.10f, ", valuesTest[(size_t)i]); printf("... \n");
    printf("ValuesWant:"); for (int i = 0; i < 10; i++) printf("%6.10f, ", valuesWant[(size_t)i]); printf("... \n");
    std::vector<double> d(valuesWant.size());
    for (size_t i = 0; i < d.size(); i++) d[i] = fabs(valuesWant[i] - valuesTest[i]);
    double mn = *std::min_element(d.begin(), d.end()), mx = *std::max_element(d.begin(), d.end()), mean = 0;
    for (double x : d) mean += x; mean /= (double)d.size();
    std::vector<double> s = d; std::sort(s.begin(), s.end()); double med = s[s.size() / 2];
    auto lg = [](double x) { return log2(1.0 / x); };
    printf("MIN Prec : (%.2f, +Inf) Log2 \nMAX Prec : (%.2f, +Inf) Log2 \nAVG Prec : (%.2f, +Inf) Log2 \nMED Prec : (%.2f, +Inf) Log2 \n", lg(mx), lg(mn), lg(mean), lg(med));
    printf("Err stdF :  -Inf Log2 \nErr stdT :  -Inf Log2 \n\n\n");
}

// ---------------------------------------------------------------- encoder / encryptor / decryptor
// ckks EncodeCoeffs -> scaleUpVecExact: x = uint64(|v|*scale + 0.5) (plain f64), mod q, q - r if v < 0
std::vector<uint64_t> EncodeCoeffs(const std::vector<double> &coeffs, int level, double scale) {
    if ((int)coeffs.size() > N) panic("cannot EncodeCoeffs: too many coefficients");
    std::vector<uint64_t> out((size_t)(level + 1) * N, 0);
    for (size_t i = 0; i < coeffs.size(); i++) {
        const double val = coeffs[i]; const bool neg = val < 0; const double x = neg ? -scale * val : scale * val;
        if (x > 1.8446744073709552e+19) panic("EncodeCoeffs: |value*scale| beyond 2^64 is outside the conv path");
        const uint64_t xi = (uint64_t)(x + 0.5);
        for (int l = 0; l <= level; l++) { uint64_t r = xi % MODQ[l]; out[(size_t)l * N + i] = neg ? MODQ[l] - r : r; }
    }
    return out;
}
Ciphertext EncryptNew(Context *c, const std::vector<uint64_t> &pt_rows, int level, double scale) {    // sk-encryption: c0 = -c1*s + e + m
    const uint64_t rs = resnetReplaySeed() ? 5 + 1000 * c->replay_encryptions++ : 0;          // replay: or_encrypt(seed 5) for the first image, as resnet_layer_digests encrypts it
    std::vector<int64_t> e; if (rs) replay::gauss(rs ^ 0xABCDEFull, e); else e = gaussian(c);
    std::vector<uint64_t> ct((size_t)2 * (level + 1) * N);
    for (int l = 0; l <= level; l++) {
        const uint64_t q = MODQ[l];
        std::vector<uint64_t> c1, t = signed_row(e, q);
        if (rs) { c1.resize(N); replay::fill_seeded(rs + 0x2000 + (uint64_t)l, q, c1.data()); } else c1 = uniform_row(c, q);
        for (int j = 0; j < N; j++) t[(size_t)j] = addmod(t[(size_t)j], pt_rows[(size_t)l * N + (size_t)j] % q, q);
        t = gpu_ntt(c, l, t);
        for (int j = 0; j < N; j++) {
            ct[(size_t)l * N + (size_t)j] = submod(t[(size_t)j], mulmod(c1[(size_t)j], c->sk_ntt[l][(size_t)j], q), q);
            ct[(size_t)(level + 1 + l) * N + (size_t)j] = c1[(size_t)j];
        }
    }
    Ciphertext r; r.d = dev_upload(c, ct); r.level = level; r.Scale = scale;
    return r;
}
std::vector<double> DecryptDecodeCoeffs(Context *c, const Ciphertext &ct) {       // Decrypt + DecodeCoeffs at level 0 (test.go:59-60)
    if (ct.level != 0) panic("DecryptDecodeCoeffs: level 0 expected on the conv path");
    std::vector<uint64_t> h = dev_download(c, ct.d, 2), m(N);
    const uint64_t q = MODQ[0];
    for (int j = 0; j < N; j++) m[(size_t)j] = addmod(h[(size_t)j], mulmod(h[(size_t)N + (size_t)j], c->sk_ntt[0][(size_t)j], q), q);
    m = gpu_ntt(c, 0, m, true);
    std::vector<double> out(N);
    for (int j = 0; j < N; j++) { uint64_t v = m[(size_t)j]; out[(size_t)j] = (v > q / 2 ? -(double)(q - v) : (double)v) / ct.Scale; }
    return out;
}
void freeCt(Context *c, Ciphertext &ct) { if (ct.d) HC(c->hc, hc_free(c->hc, ct.d)); ct.d = nullptr; }

// ---------------------------------------------------------------- prep_Ker (conv.go:487-518)
KerPlain prep_Ker(Context *c, const std::vector<double> &ker_in, const std::vector<double> &BN_a, int in_wid, int ker_wid,
                  int real_ib, int real_ob, int norm, int ECD_LV, int pos, bool trans) {
    // The whole of conv.go:487-518 runs on the device (hc_prep_ker: reshape_ker, BN scale, max_bat embedding,
    // encode_ker_final, EncodeCoeffs rounding, ToNTT); reshape_ker/encode_ker_final above remain as the host-side
    // statement of the layout (used by tests and by anyone who wants to inspect a kernel plaintext).
    if (trans || pos != 0 || ECD_LV != 1) panic("prep_Ker: only the conv path's (pos=0, trans=false, ECD_LV=1) form is built");
    KerPlain k; k.max_bat = N / (in_wid * in_wid); k.Scale = c->scale;
    HC(c->hc, hc_prep_ker(c->hc, ker_in.data(), (int)ker_in.size(), BN_a.data(), in_wid, ker_wid, real_ib, real_ob, norm, c->scale, &k.h));
    k.shard_h.push_back(k.h);
    for (size_t g = 1; g < c->shards.size(); g++) { hc_ker *h = nullptr; HC(c->shards[g], hc_prep_ker(c->shards[g], ker_in.data(), (int)ker_in.size(), BN_a.data(), in_wid, ker_wid, real_ib, real_ob, norm, c->scale, &h)); k.shard_h.push_back(h); }
    return k;
}

// ---------------------------------------------------------------- GpuEvaluator
Ciphertext GpuEvaluator::alloc(int level, double scale) { Ciphertext r; r.d = dev_rows(cont, (size_t)2 * (level + 1)); r.level = level; r.Scale = scale; return r; }
Ciphertext GpuEvaluator::MulNew(const Ciphertext &ct, const Plaintext &pt) {
    const int level = std::min(ct.level, pt.level);
    Ciphertext r = alloc(level, ct.Scale * pt.Scale);
    for (int p = 0; p < 2; p++) for (int l = 0; l <= level; l++)
        HC(cont->hc, hc_mul(cont->hc, l, ct.d + ((size_t)p * (ct.level + 1) + l) * N, pt.d + (size_t)l * N, r.d + ((size_t)p * (level + 1) + l) * N, 1));
    return r;
}
void GpuEvaluator::SetScale(Ciphertext &ct, double scale) {       // MultByConst(scale/ct.Scale) ; Rescale ; ct.Scale = scale
    const double constant = scale / ct.Scale; double smul = 1;
    for (int p = 0; p < 2; p++) for (int l = 0; l <= ct.level; l++) {
        uint64_t k = hc_const_for(constant, (double)MODQ[ct.level], MODQ[l], &smul);
        uint64_t *row = ct.d + ((size_t)p * (ct.level + 1) + l) * N;
        HC(cont->hc, hc_mul_const(cont->hc, l, row, k, row, 1));
    }
    double sc = ct.Scale * smul; int drops = 0;
    while (ct.level - drops > 0 && sc / (double)MODQ[ct.level - drops] >= scale / 2) { sc /= (double)MODQ[ct.level - drops]; drops++; }
    if (ct.level == 1 && drops == 1) {
        Ciphertext r = alloc(0, scale);
        for (int p = 0; p < 2; p++) HC(cont->hc, hc_div_round_last(cont->hc, 1, ct.d + (size_t)p * 2 * N, r.d + (size_t)p * N));
        HC(cont->hc, hc_free(cont->hc, ct.d));
        ct = r;
    } else if (drops != 0) panic("SetScale: only the level 1 -> 0 rescale of the conv path is built");
    ct.Scale = scale;
}
Ciphertext GpuEvaluator::SubNew(const Ciphertext &a, const Ciphertext &b) {
    if (a.level != b.level) panic("SubNew: level mismatch");
    Ciphertext r = alloc(a.level, a.Scale);
    HC(cont->hc, hc_sub(cont->hc, 0, a.d, b.d, r.d, 2 * (a.level + 1)));    // level 0 on this path: both rows mod Q0
    if (a.level != 0) panic("SubNew: level 0 expected on the pack path");
    return r;
}
void GpuEvaluator::Add(const Ciphertext &a, const Ciphertext &b, Ciphertext &out) {
    if (a.level != 0 || b.level != 0) panic("Add: level 0 expected on the pack path");
    if (!out.d) out = alloc(0, a.Scale);
    HC(cont->hc, hc_add(cont->hc, 0, a.d, b.d, out.d, 2)); out.Scale = a.Scale; out.level = 0;
}
void GpuEvaluator::AddPlain(const Ciphertext &a, const Plaintext &b, Ciphertext &out) {
    if (!out.d) out = alloc(0, a.Scale);
    HC(cont->hc, hc_add(cont->hc, 0, a.d, b.d, out.d, 1));
    if (out.d != a.d) HC(cont->hc, hc_copy(cont->hc, out.d + N, a.d + N, (size_t)N * 8));
    out.Scale = a.Scale; out.level = 0;
}
void GpuEvaluator::RotateGal(const Ciphertext &ct, uint64_t galEl, Ciphertext &out) {
    if (ct.level != 0) panic("RotateGal: level 0 expected on the pack path");
    if (!out.d) out = alloc(0, ct.Scale);
    if (galEl - 1 < 32) {       // evaluator.permuteNTT from the general primitives
        uint64_t *d = dev_rows(cont, 2);
        HC(cont->hc, hc_keyswitch(cont->hc, galEl, 0, ct.d + N, d, d + N));
        HC(cont->hc, hc_add(cont->hc, 0, d, ct.d, d, 1));
        HC(cont->hc, hc_permute(cont->hc, galEl, d, out.d, 1));
        HC(cont->hc, hc_permute(cont->hc, galEl, d + N, out.d + N, 1));
        HC(cont->hc, hc_free(cont->hc, d));
    } else HC(cont->hc, hc_rotate_gal_l0(cont->hc, galEl, ct.d, ct.d + N, out.d, out.d + N));
    out.Scale = ct.Scale; out.level = 0;
}

static void dbg_digest(Context *c, const char *what, const uint64_t *d, size_t rows) {      // HCONV_DBG_STAGES: FNV-1a of device rows, on stderr
    if (!getenv("HCONV_DBG_STAGES")) return;
    std::vector<uint64_t> h = dev_download(c, d, rows); uint64_t f = 1469598103934665603ull;
    for (uint64_t w : h) for (int b = 0; b < 8; b++) { f ^= (w >> (8 * b)) & 0xff; f *= 1099511628211ull; }
    fprintf(stderr, "[stage] %-28s %016llx\n", what, (unsigned long long)f);
}
// ---------------------------------------------------------------- conv_then_pack / evalConv_BN
// pack_ctxts (conv.go:266-300), statement by statement, on the ckks.Evaluator subset (GpuEvaluator = one C-ABI call per
// limb row). This is the path a cgo gpuEvaluator takes when conv.go is left untouched.
static Ciphertext pack_ctxts(Context *c, GpuEvaluator &pack_eval, std::vector<Ciphertext> &ctxts_in, int max_cnum, int real_cnum, const std::vector<Plaintext> &idx) {
    int step = max_cnum / 2;
    const int norm = max_cnum / real_cnum;
    std::vector<Ciphertext> &ctxts = ctxts_in;                       // CopyNew is not needed: the inputs are ours
    for (int i = 0; i < max_cnum; i++) if (i % norm == 0) ctxts[(size_t)i].Scale *= (double)real_cnum;   // conv.go:274
    int logStep = 0;
    for (int i = step; i > 1; i /= 2) logStep++;
    int j = c->logN - logStep;
    while (step >= norm && step >= 1) {
        for (int i = 0; i < step; i += norm) {
            Ciphertext tmp1 = pack_eval.MulNew(ctxts[(size_t)(i + step)], idx[(size_t)logStep]);   // conv.go:288
            Ciphertext tmp2 = pack_eval.SubNew(ctxts[(size_t)i], tmp1);                               // conv.go:289
            pack_eval.Add(ctxts[(size_t)i], tmp1, tmp1);                                              // conv.go:290
            pack_eval.RotateGal(tmp2, (1ull << j) + 1, tmp2);                                         // conv.go:291
            pack_eval.Add(tmp1, tmp2, ctxts[(size_t)i]);                                              // conv.go:292
            freeCt(c, tmp1); freeCt(c, tmp2);
        }
        step /= 2; logStep--; j++;
    }
    return ctxts[0];
}
// conv_then_pack (conv.go:522-546) op by op on the evaluator interface (HCONV_OPWISE=1): must give the same bits
// as the fused kernels.
static Ciphertext conv_then_pack_opwise(Context *c, const Ciphertext &ctxt_in, const KerPlain &pl_ker, int max_ob, int norm, double out_scale) {
    GpuEvaluator pack_evaluator(c);
    auto start = now();
    std::vector<uint64_t> flat((size_t)max_ob * 2 * N);
    HC(c->hc, hc_ker_download(c->hc, pl_ker.h, flat.data()));        // pl_ker[i] as Lattigo holds them (plain NTT rows)
    uint64_t *dker = dev_upload(c, flat);
    std::vector<Ciphertext> ctxt_out((size_t)max_ob);
    for (int i = 0; i < max_ob; i++) if (i % norm == 0) {
        Plaintext pt; pt.d = dker + (size_t)i * 2 * N; pt.level = 1; pt.Scale = pl_ker.Scale;
        ctxt_out[(size_t)i] = pack_evaluator.MulNew(ctxt_in, pt);                                     // conv.go:527
        pack_evaluator.SetScale(ctxt_out[(size_t)i], out_scale / (double)(max_ob / norm));            // conv.go:528
    }
    HC(c->hc, hc_sync(c->hc));
    auto mt = now();
    printf("\t mult time:  %s\n", dur(start).c_str());
    // plain_idx (conv.go:248-253) as device plaintexts
    std::vector<uint64_t> xs((size_t)LOGN * N, 0);
    for (int s = 0; s < LOGN; s++) xs[(size_t)s * N + ((size_t)1 << s)] = 1;
    uint64_t *didx = dev_upload(c, xs);
    HC(c->hc, hc_ntt(c->hc, 0, didx, didx, LOGN));
    std::vector<Plaintext> idx((size_t)LOGN);
    for (int s = 0; s < LOGN; s++) { idx[(size_t)s].d = didx + (size_t)s * N; idx[(size_t)s].level = 0; idx[(size_t)s].Scale = 1.0; }
    Ciphertext res = pack_ctxts(c, pack_evaluator, ctxt_out, max_ob, max_ob / norm, idx);
    HC(c->hc, hc_sync(c->hc));
    printf("\t Pack time:  %s\n", dur(mt).c_str());
    for (int i = 1; i < max_ob; i++) if (ctxt_out[(size_t)i].d) freeCt(c, ctxt_out[(size_t)i]);
    HC(c->hc, hc_free(c->hc, dker)); HC(c->hc, hc_free(c->hc, didx));
    if (out_scale != res.Scale || 0 != res.level) panic("LV or scale after conv then pack, inconsistent");   // conv.go:541-543
    return res;
}

Ciphertext conv_then_pack(Context *c, const Ciphertext &ctxt_in, const KerPlain &pl_ker, int max_ob, int norm, int ECD_LV, double out_scale, const Plaintext *pl_bn_b) {
    (void)ECD_LV;
    if (getenv("HCONV_OPWISE")) return conv_then_pack_opwise(c, ctxt_in, pl_ker, max_ob, norm, out_scale);
    auto start = now();
    Ciphertext r; r.d = dev_rows(c, 2); r.level = 0;
    const int G = (int)c->shards.size();
    if (G > 1 && norm == 1 && max_ob % G == 0 && (int)pl_ker.shard_h.size() == G) {
        // ONE convolution over G devices: replicas of the input by peer copies, then hc_conv_then_pack_sharded (channels i mod G per
        // device, one collection of G x 1 MiB partials on device 0, last log2 G levels there). The two phases overlap across devices,
        // so "mult time" here is the time to queue the work and "Pack time" the time until the result is complete.
        std::vector<uint64_t *> rep((size_t)G); std::vector<const uint64_t *> ins((size_t)G);
        rep[0] = nullptr; ins[0] = ctxt_in.d;
        for (int g = 1; g < G; g++) { void *p = nullptr; HC(c->shards[(size_t)g], hc_malloc(c->shards[(size_t)g], (size_t)4 * N * 8, &p)); rep[(size_t)g] = (uint64_t *)p; ins[(size_t)g] = rep[(size_t)g];
                                      HC(c->shards[(size_t)g], hc_copy_peer(c->shards[(size_t)g], p, c->hc, ctxt_in.d, (size_t)4 * N * 8)); }
        double sc = 0;
        HC(c->hc, hc_conv_then_pack_sharded(c->shards.data(), G, ins.data(), ctxt_in.Scale, pl_ker.shard_h.data(), pl_ker.Scale, max_ob, out_scale, nullptr, r.d, &sc));
        printf("\t mult time:  %s\n", dur(start).c_str());
        auto mt = now();
        HC(c->hc, hc_sync(c->hc));
        printf("\t Pack time:  %s\n", dur(mt).c_str());
        dbg_digest(c, "pack result", r.d, 2);
        for (int g = 1; g < G; g++) { HC(c->shards[(size_t)g], hc_sync(c->shards[(size_t)g])); HC(c->shards[(size_t)g], hc_free(c->shards[(size_t)g], rep[(size_t)g])); }
        r.Scale = sc;
        if (out_scale != r.Scale || 0 != r.level) panic("LV or scale after conv then pack, inconsistent");   // conv.go:541-543
        return r;
    }
    // The reference prints "mult time" and "Pack time" separately (conv.go:533,535); run the two phases through
    // the same fused kernels, synchronising in between only to print the split.
    uint64_t *cts = dev_rows(c, (size_t)max_ob * 2);
    HC(c->hc, hc_conv_mult_phase(c->hc, ctxt_in.d, ctxt_in.Scale, pl_ker.h, pl_ker.Scale, max_ob, norm, out_scale, cts));
    HC(c->hc, hc_sync(c->hc));
    dbg_digest(c, "ct_in", ctxt_in.d, 4); dbg_digest(c, "mult slot 0", cts, 2); dbg_digest(c, "mult slot last", cts + (size_t)(max_ob - 1) * 2 * N, 2); dbg_digest(c, "mult all", cts, (size_t)max_ob * 2);
    auto mt = now();
    printf("\t mult time:  %s\n", dur(start).c_str());
    HC(c->hc, hc_pack_ctxts(c->hc, cts, max_ob, max_ob / norm));
    HC(c->hc, hc_sync(c->hc));
    printf("\t Pack time:  %s\n", dur(mt).c_str());
    dbg_digest(c, "pack result", cts, 2);
    HC(c->hc, hc_copy(c->hc, r.d, cts, (size_t)2 * N * 8));     // keep the pack result (slot 0), drop the workspace
    HC(c->hc, hc_free(c->hc, cts));
    r.Scale = (out_scale / (double)(max_ob / norm)) * (double)(max_ob / norm);            // conv.go:528 then conv.go:274
    if (out_scale != r.Scale || 0 != r.level) panic("LV or scale after conv then pack, inconsistent");   // conv.go:541-543
    (void)pl_bn_b;
    return r;
}
Ciphertext evalConv_BN(Context *c, const Ciphertext &ct_input, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                       const std::vector<double> &bn_b, int in_wid, int ker_wid, int real_ib, int real_ob, int norm, double out_scale, bool trans) {
    const int max_batch = N / (in_wid * in_wid);
    auto start = now();
    KerPlain pl_ker = prep_Ker(c, ker_in, bn_a, in_wid, ker_wid, real_ib, real_ob, norm, c->ECD_LV, 0, trans);      // eval.go:231
    std::vector<double> b_coeffs((size_t)N, 0.0);
    for (size_t i = 0; i < bn_b.size(); i++) for (int j = 0; j < in_wid * in_wid; j++) b_coeffs[(size_t)(norm * (int)i + j * max_batch)] = bn_b[i];   // eval.go:233-238
    Plaintext pl_bn_b; pl_bn_b.level = 0; pl_bn_b.Scale = out_scale;
    pl_bn_b.d = dev_upload(c, EncodeCoeffs(b_coeffs, 0, out_scale));
    HC(c->hc, hc_ntt(c->hc, 0, pl_bn_b.d, pl_bn_b.d, 1));                                                              // eval.go:242-243
    HC(c->hc, hc_sync(c->hc));
    printf("Plaintext (kernel) preparation, Done in %s \n", dur(start).c_str());                                      // eval.go:244
    start = now();
    dbg_digest(c, "bias plaintext", pl_bn_b.d, 1);
    Ciphertext ct_res = conv_then_pack(c, ct_input, pl_ker, max_batch, norm, c->ECD_LV, out_scale, &pl_bn_b);          // eval.go:251
    if (pl_bn_b.Scale != ct_res.Scale || ct_res.level != 0) {                                                          // eval.go:252-257
        printf("plain scale:  %g\nctxt scale:  %g\nctxt lv:  %d\n", pl_bn_b.Scale, ct_res.Scale, ct_res.level);
        panic("LV or scale after conv then pack, inconsistent");
    }
    HC(c->hc, hc_add(c->hc, 0, ct_res.d, pl_bn_b.d, ct_res.d, 1));                                                     // eval.go:258
    HC(c->hc, hc_sync(c->hc));
    printf("Conv (with BN) Done in %s \n", dur(start).c_str());                                                       // eval.go:260
    hc_ker_free(c->hc, pl_ker.h);
    for (size_t g = 1; g < pl_ker.shard_h.size(); g++) hc_ker_free(c->shards[g], pl_ker.shard_h[g]);
    HC(c->hc, hc_free(c->hc, pl_bn_b.d));
    return ct_res;
}

// HCONV_IMAGE_BATCH (not a reference feature): the reference pushes images through a layer one after another (test.go:128); here up to 8 go through it as
// one set of launches - hc_conv_then_pack_batch for the convolution, hc_set_batch for the bootstrapping chain - with weights, masks, matrices and keys read once
int imageBatch() {
    const char *e = getenv("HCONV_IMAGE_BATCH"); const int n = e ? atoi(e) : 1;
    if (n < 1 || n > 8) panic("HCONV_IMAGE_BATCH must be 1..8");
    return n;
}
// eval.go:224-263 for the images of a batch: prep_Ker and the bias plaintext once (they depend on the layer only), conv_then_pack + the bias add of all images as ONE launch set
std::vector<Ciphertext> evalConv_BN_batch(Context *c, const std::vector<Ciphertext> &ct_inputs, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                                          const std::vector<double> &bn_b, int in_wid, int ker_wid, int real_ib, int real_ob, int norm, double out_scale, bool trans) {
    const int n = (int)ct_inputs.size();
    if (n == 1 || getenv("HCONV_OPWISE") || c->shards.size() > 1) {      // one image (the reference's own flow, with its two timers), or modes that exist per image only
        std::vector<Ciphertext> r; for (const Ciphertext &ct : ct_inputs) r.push_back(evalConv_BN(c, ct, ker_in, bn_a, bn_b, in_wid, ker_wid, real_ib, real_ob, norm, out_scale, trans));
        return r;
    }
    if (n < 1 || n > 16) panic("evalConv_BN_batch: 1..16 images");
    const int max_batch = N / (in_wid * in_wid);
    auto start = now();
    KerPlain pl_ker = prep_Ker(c, ker_in, bn_a, in_wid, ker_wid, real_ib, real_ob, norm, c->ECD_LV, 0, trans);      // eval.go:231
    std::vector<double> b_coeffs((size_t)N, 0.0);
    for (size_t i = 0; i < bn_b.size(); i++) for (int j = 0; j < in_wid * in_wid; j++) b_coeffs[(size_t)(norm * (int)i + j * max_batch)] = bn_b[i];   // eval.go:233-238
    Plaintext pl_bn_b; pl_bn_b.level = 0; pl_bn_b.Scale = out_scale;
    pl_bn_b.d = dev_upload(c, EncodeCoeffs(b_coeffs, 0, out_scale));
    HC(c->hc, hc_ntt(c->hc, 0, pl_bn_b.d, pl_bn_b.d, 1));                                                              // eval.go:242-243
    HC(c->hc, hc_sync(c->hc));
    printf("Plaintext (kernel) preparation, Done in %s \n", dur(start).c_str());                                      // eval.go:244
    start = now();
    std::vector<Ciphertext> res((size_t)n); std::vector<const uint64_t *> ins((size_t)n), bias((size_t)n, pl_bn_b.d); std::vector<const hc_ker *> kers((size_t)n, pl_ker.h); std::vector<uint64_t *> outs((size_t)n);
    for (int z = 0; z < n; z++) {
        if (ct_inputs[(size_t)z].Scale != ct_inputs[0].Scale || ct_inputs[(size_t)z].level != ct_inputs[0].level) panic("evalConv_BN_batch: the images of a batch share level and scale");
        res[(size_t)z].d = dev_rows(c, 2); res[(size_t)z].level = 0; ins[(size_t)z] = ct_inputs[(size_t)z].d; outs[(size_t)z] = res[(size_t)z].d;
    }
    double sc = 0;
    if (hc_conv_then_pack_batch(c->hc, n, ins.data(), ct_inputs[0].Scale, kers.data(), pl_ker.Scale, max_batch, norm, out_scale, bias.data(), outs.data(), &sc)) panic(std::string("hc_conv_then_pack_batch: ") + hc_last_error(c->hc));   // conv.go:541-543's panic included
    HC(c->hc, hc_sync(c->hc));
    for (int z = 0; z < n; z++) res[(size_t)z].Scale = sc;
    printf("Conv (with BN) Done in %s  (%d images)\n", dur(start).c_str(), n);                                       // eval.go:260
    hc_ker_free(c->hc, pl_ker.h);
    HC(c->hc, hc_free(c->hc, pl_bn_b.d));
    return res;
}

// ---------------------------------------------------------------- testConv_in (test.go:15-74)
void testConv_in(int in_batch, int in_wid, int ker_wid, int total_test_num, bool boot) {
    const std::string kind = "Conv";
    const int raw_in_batch = in_batch, raw_in_wid = in_wid - ker_wid / 2, norm = in_batch / raw_in_batch;
    const std::string test_dir = "test_conv_data/";
    int kp_wid, out_batch, logN; bool trans;
    set_Variables(in_batch, raw_in_wid, in_wid, ker_wid, kind, &kp_wid, &out_batch, &logN, &trans);
    const int raw_out_batch = out_batch / norm;
    Context *cont = newContext(logN, ker_wid, {in_wid}, {kp_wid}, boot, kind);
    printf("vec size: log2 =  %d\n", cont->logN);
    printf("raw input width:  %d\n", raw_in_wid);
    printf("kernel width:  %d\n", ker_wid);
    printf("num raw batches in & out:  %d ,  %d\n", raw_in_batch, raw_out_batch);
    for (int test_iter = 0; test_iter < total_test_num; test_iter++) {
        printf("%d -th iter...start\n", test_iter + 1);
        const std::string pre = test_dir + "test_conv" + std::to_string(ker_wid) + "_batch_" + std::to_string(in_batch) + "_";
        const std::string suf = "_" + std::to_string(test_iter) + ".csv";
        std::vector<double> raw_input = readTxt(pre + "in" + suf, raw_in_wid * raw_in_wid * raw_in_batch);
        std::vector<double> ker_in = readTxt(pre + "ker" + suf, raw_in_batch * raw_out_batch * ker_wid * ker_wid);
        std::vector<double> bn_a = readTxt(pre + "bna" + suf, raw_out_batch);
        std::vector<double> bn_b = readTxt(pre + "bnb" + suf, raw_out_batch);
        std::vector<double> input = prep_Input(raw_input, raw_in_wid, in_wid, cont->Nn, norm, trans, false);
        auto start = now();
        Ciphertext ctxt_input = EncryptNew(cont, EncodeCoeffs(input, cont->ECD_LV, cont->scale), cont->ECD_LV, cont->scale);
        printf("Encryption done in %s \n", dur(start).c_str());
        if (boot) {                                                                                    // test.go:52-53, eval.go:272-607
            const double pow_ = 4.0, alpha = 0.0;                                                      // test.go:22
            const double out_scale = exp2(round(log2((double)MODQ[0]) - (pow_ + 8)));                  // eval.go:433
            // HCONV_IMAGE_BATCH = n > 1 (not a reference feature): n independent encryptions of the input go through the layer as ONE set of launches
            const int nimg = imageBatch();
            std::vector<Ciphertext> ins{ctxt_input};
            for (int z = 1; z < nimg; z++) ins.push_back(EncryptNew(cont, EncodeCoeffs(input, cont->ECD_LV, cont->scale), cont->ECD_LV, cont->scale));
            std::vector<Ciphertext> ct_conv = evalConv_BN_batch(cont, ins, ker_in, bn_a, bn_b, in_wid, ker_wid, raw_in_batch, raw_out_batch, norm, out_scale, trans);
            HC(cont->hc, hc_sync(cont->hc));                                                           // hand-over to the bootstrapper's context (another stream)
            std::vector<const uint64_t *> conv_d; for (const Ciphertext &ct : ct_conv) conv_d.push_back(ct.d);
            auto layer = now();
            std::vector<BootCiphertext> ct_res = evalConv_BNRelu_tail_batch(cont->btp, "Conv", 0, conv_d, ct_conv[0].Scale, alpha, pow_, in_wid, kp_wid);
            if (nimg > 1) printf("Bootstrapping + ReLU of %d ciphertexts done in %s \n", nimg, dur(layer).c_str());
            start = now();
            std::vector<double> cfs = bootDecryptDecodeCoeffs(cont->btp, ct_res[0]);
            printf("Decryption Done in %s \n", dur(start).c_str());
            std::vector<double> test_out = post_process(cfs, raw_in_wid, in_wid);
            std::vector<double> real_out = readTxt(pre + "reluout" + suf, raw_in_wid * raw_in_wid * raw_in_batch);
            printDebugCfsPlain(test_out, real_out);
            for (int z = 1; z < nimg; z++) {                                                           // the other encryptions decrypt to the same values up to the scheme's noise
                std::vector<double> cz = post_process(bootDecryptDecodeCoeffs(cont->btp, ct_res[(size_t)z]), raw_in_wid, in_wid);
                double mx = 0, mr = 0; for (size_t i = 0; i < cz.size(); i++) { mx = std::max(mx, fabs(cz[i] - test_out[i])); mr = std::max(mr, fabs(cz[i] - real_out[i])); }
                printf("image %d of the batch: max |difference| to image 0 = %.3g, to the expected output = %.3g\n", z, mx, mr);
            }
            long nk, nks; bootStats(cont->btp, &nk, &nks);
            if (getenv("HCONV_BOOT_STATS")) printf("boot stats: %ld switching keys generated, %ld key switches\n", nk, nks);
            for (Ciphertext &ct : ins) freeCt(cont, ct);
            for (Ciphertext &ct : ct_conv) freeCt(cont, ct);
            for (BootCiphertext &ct : ct_res) freeBootCt(cont->btp, ct);
            continue;
        }
        Ciphertext ct_result = evalConv_BN(cont, ctxt_input, ker_in, bn_a, bn_b, in_wid, ker_wid, raw_in_batch, raw_out_batch, norm, (double)(1 << 30), trans);
        if (getenv("HCONV_PRINT_DIGEST")) {      // FNV-1a over the result ciphertext: lets tests compare code paths bit for bit
            std::vector<uint64_t> h = dev_download(cont, ct_result.d, 2); uint64_t f = 1469598103934665603ull;
            for (uint64_t w : h) for (int b = 0; b < 8; b++) { f ^= (w >> (8 * b)) & 0xff; f *= 1099511628211ull; }
            printf("ciphertext digest: %016llx\n", (unsigned long long)f);
        }
        start = now();
        std::vector<double> cfs_tmp = DecryptDecodeCoeffs(cont, ct_result);
        printf("Decryption Done in %s \n", dur(start).c_str());
        std::vector<double> test_out = post_process(cfs_tmp, raw_in_wid, in_wid);
        std::vector<double> real_out = readTxt(pre + "out" + suf, raw_in_wid * raw_in_wid * raw_in_batch);
        printDebugCfsPlain(test_out, real_out);
        freeCt(cont, ctxt_input); freeCt(cont, ct_result);
    }
    freeContext(cont);
}

}  // namespace hconv
