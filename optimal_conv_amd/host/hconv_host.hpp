// hconv_host.hpp — C++ host side above the C ABI (include/hconv.h), mirroring the reference's operator interface
// for the `conv` path with the same names, argument meaning and error behaviour (errors are fatal, like Go's panic).
//
// The reference's host code is Go; no Go toolchain exists in this image or on the GPU box,
// so per the build contract the host side is C++. Mapping (reference -> here):
//   main.go:22-42   type context              -> struct Context
//   main.go:44-462  newContext("Conv")        -> newContext()
//   conv.go:241-261 gen_idxNlogs              -> inside newContext (idx on the device, Galois keys 2^(i+1)+1)
//   main.go:1007    prep_Input                -> prep_Input
//   conv.go:184     reshape_ker               -> reshape_ker
//   conv.go:206     encode_ker_final          -> encode_ker_final
//   conv.go:487     prep_Ker                  -> prep_Ker      (EncodeCoeffs on the host, ToNTT on the GPU)
//   conv.go:522     conv_then_pack            -> conv_then_pack (fused: hc_conv_then_pack; or op-by-op through
//   conv.go:266     pack_ctxts                -> pack_ctxts      GpuEvaluator with HCONV_OPWISE=1)
//   eval.go:224     evalConv_BN               -> evalConv_BN
//   main.go:1057    post_process              -> post_process
//   main.go:971     readTxt                   -> readTxt
//   main.go:694     printDebugCfsPlain        -> printDebugCfsPlain
//   test.go:15      testConv_in               -> testConv_in
// ckks.Evaluator methods the path calls -> class GpuEvaluator {MulNew, SetScale, SubNew, Add, RotateGal}.
// Keys/encryption/decryption are harness-only (the reference draws them from crypto-random; nothing to match):
// implemented here on the host with every NTT done on the GPU through the ABI.
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/hconv.h"
#include "../../include/hconv_test_hooks.h"      // --test-mode replays only (HCONV_RESNET_REPLAY)
#include "hconv_prng.hpp"

namespace hconv {

typedef unsigned __int128 u128;
static const int LOGN = 16;
static const int N = 1 << LOGN;

// ckks.DefaultBootstrapParams[6] of the fork, level order; only Q[0], Q[1] carry the conv path
extern const std::vector<uint64_t> PARAMS6_Q;
extern const std::vector<uint64_t> PARAMS7_Q;      // ckks.DefaultBootstrapParams[7]: the baseline's chain (main.go:54)
extern const std::vector<uint64_t> PARAMS6_P;      // bootstrapping key-switch primes (logQP print only)
static const uint64_t PACK_P = 0x1fffffffffe00001ull;   // main.go:449

[[noreturn]] void panic(const std::string &msg);        // Go's panic(): message to stderr, exit status 2
// The library reads no configuration from the environment (include/hconv.h); this CLI does, and hands it over as options right after every hc_ctx_create:
// HCONV_ASYNC_ALLOC (0 / 1 / 2), HCONV_SMALL32, HCONV_ROT_FUSE, HCONV_PACK32 (A/B switches; same residues in every setting). pack32_default: the value a context takes
// when HCONV_PACK32 is unset (-1: the library's own default)
void applyEnvOptions(hc_ctx *hc, int pack32_default = -1);
// Test-only overrides (HCONV_SEED: deterministic keys; HCONV_CHAIN_REPLAY: planted keys and input) take effect only when the CLI was
// started with --test-mode as its first argument; set in the environment WITHOUT that flag they end the process (an inherited variable
// must never turn a deployment into key-less or predictable computation).
bool &testMode();
const char *testOnlyEnv(const char *name);              // getenv(name) under --test-mode; nullptr when unset; panic when set without the flag
// HCONV_RESNET_REPLAY=<seed> (test mode): the secret key, every switching key and the input's encryption randomness are the counter-based splitmix64 draws of the
// test oracle's harness generators (or_gen_sk / or_gen_galois_key_l0 / or_gen_swk / or_encrypt, restated in hconv_host.cpp): the `resnet` run then
// computes, bit for bit, the network tests/golden/gen_resnet_digests.py ran on the oracle, and prints a `replay digest layer i` line per layer (tests/test_gpu_z_cli.py)
uint64_t resnetReplaySeed();                            // 0: off
namespace replay {                                      // the oracle harness' generators (splitmix64, Box-Muller sigma 3.2 bound 6 sigma)
uint64_t sm64(uint64_t seed, uint64_t i);
void fill_seeded(uint64_t seed, uint64_t q, uint64_t *out);
void gauss(uint64_t seed, std::vector<int64_t> &e);
std::vector<int64_t> gen_sk(uint64_t seed, int h);
}

// Device-resident ciphertext / plaintext (ckks.Ciphertext{Value []*ring.Poly; Scale}, ckks.Plaintext)
struct Ciphertext {
    uint64_t *d = nullptr;   // device [2][level+1][N]
    int level = 0;
    double Scale = 0;
};
struct Plaintext {
    uint64_t *d = nullptr;   // device [level+1][N], NTT domain
    int level = 0;
    double Scale = 0;
};

struct Boot;                                   // hconv_relu.cpp: bootstrapper + leveled evaluator (cont.btp, cont.evaluator of main.go:464-507)
struct BootCiphertext { uint64_t *d = nullptr; int level = 0; double Scale = 0; };   // device [2][level+1][N]

struct Context {
    int logN = LOGN, Nn = N, ECD_LV = 1;       // main.go:46
    Boot *btp = nullptr;                      // only when newContext(..., boot = true)
    hc_ctx *hc = nullptr;                     // pack_evaluator + evaluator (both run on the same device context)
    std::vector<hc_ctx *> shards;             // HCONV_GPUS=G > 1: shards[0] = hc, shards[g] = a context on device g (mod the device count)
                                              // with the same Galois keys: ONE convolution runs i mod G over them (`conv 7 3`)
    std::vector<int64_t> sk;                  // sparse ternary secret, h = 192 (main.go:410)
    std::vector<uint64_t> sk_ntt[3];          // NTT rows mod Q0, Q1, P (host)
    double scale = (double)(1 << 30);
    int num_rotations = 0;
    Seed256 seed;                             // key of this context's generators (hconv_prng.hpp)
    ChaChaRng g;                              // this context's own generator: secret key, Galois keys, encryption randomness
    uint64_t replay_encryptions = 0;          // HCONV_RESNET_REPLAY: encryptions so far (the i-th takes the oracle harness' seed 5 + 1000 i)
};

// ---- harness / reference-shaped API ----
Context *newContext(int logN, int ker_wid, const std::vector<int> &in_wids, const std::vector<int> &kp_wids, bool boot, const std::string &kind);
void freeContext(Context *);
std::vector<double> readTxt(const std::string &name_file, int size);
std::vector<double> prep_Input(const std::vector<double> &input, int raw_in_wid, int in_wid, int Nn, int norm, bool trans, bool printResult);
std::vector<std::vector<double>> reshape_ker(const std::vector<double> &ker_in, int k_sz, int out_batch, bool trans);
std::vector<double> encode_ker_final(const std::vector<std::vector<double>> &ker_in, int pos, int i, int in_wid, int in_batch, int ker_wid);
std::vector<double> post_process(const std::vector<double> &in_cfs, int raw_in_wid, int in_wid);
void printDebugCfsPlain(const std::vector<double> &valuesTest, const std::vector<double> &valuesWant);
void set_Variables(int batch, int raw_in_wid, int in_wid, int ker_wid, const std::string &kind, int *kp_wid, int *out_batch, int *logN, bool *trans);

// encoder / encryptor / decryptor (ckks.Encoder.EncodeCoeffs, Encryptor.EncryptNew, Decryptor.Decrypt + DecodeCoeffs)
std::vector<uint64_t> EncodeCoeffs(const std::vector<double> &coeffs, int level, double scale);   // host rows, coefficient domain
Ciphertext EncryptNew(Context *cont, const std::vector<uint64_t> &pt_rows, int level, double scale);
std::vector<double> DecryptDecodeCoeffs(Context *cont, const Ciphertext &ct);
void freeCt(Context *cont, Ciphertext &ct);

// kernel plaintexts: handle to the B device-resident plaintexts
struct KerPlain { hc_ker *h = nullptr; int max_bat = 0; double Scale = 0; std::vector<hc_ker *> shard_h; };   // shard_h[g]: the same plaintexts on device g
KerPlain prep_Ker(Context *cont, const std::vector<double> &ker_in, const std::vector<double> &BN_a, int in_wid, int ker_wid,
                  int real_ib, int real_ob, int norm, int ECD_LV, int pos, bool trans);

Ciphertext conv_then_pack(Context *cont, const Ciphertext &ctxt_in, const KerPlain &pl_ker, int max_ob, int norm, int ECD_LV,
                          double out_scale, const Plaintext *pl_bn_b);
Ciphertext evalConv_BN(Context *cont, const Ciphertext &ct_input, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                       const std::vector<double> &bn_b, int in_wid, int ker_wid, int real_ib, int real_ob, int norm, double out_scale, bool trans);
void testConv_in(int in_batch, int in_wid, int ker_wid, int total_test_num, bool boot);
// ---- convReLU chain (hconv_relu.cpp; eval.go:272-607 for kind "Conv") ----
Boot *newBoot(const std::vector<int64_t> &sk, const Seed256 &seed, int device, const std::vector<int> &log_sparse_sets, int image_batch = 1);   // one "bootstrapper" per log_sparse; image_batch: images per launch set of the tail (HCONV_IMAGE_BATCH)
void freeBoot(Boot *);
void bootPrepareCompress(Boot *B, int in_wid, int kp_wid, int log_sparse);   // rotation keys of a stride layer's ext_double_ctxt
// everything after evalConv_BN: Scale *= 2^pow, BootstrappConv_CtoS, evalReLU + MulByPow2, keep_ctxt, BootstrappConv_StoC
BootCiphertext evalConv_BNRelu_tail(Boot *B, const std::string &kind, int log_sparse, const uint64_t *ct_conv_dev, double ct_scale, double alpha, double pow, int in_wid, int kp_wid);
// the same tail for the images of a batch (same layer, same weights) as ONE set of launches; at most the image_batch the bootstrapper was built with
std::vector<BootCiphertext> evalConv_BNRelu_tail_batch(Boot *B, const std::string &kind, int log_sparse, const std::vector<const uint64_t *> &ct_conv_dev, double ct_scale, double alpha, double pow, int in_wid, int kp_wid);
std::vector<double> bootDecryptDecodeCoeffs(Boot *B, const BootCiphertext &ct);
void freeBootCt(Boot *B, BootCiphertext &ct);
void bootStats(Boot *B, long *keys, long *keyswitches);
// ---- the baseline's bootstrapping + ReLU (test_BL.go:113-168) on parameter set [7]: cont.btp.Bootstrapp (the stock full-slot
// bootstrapper), imaginary packing / unpacking, evalReLU + MulByPow2 + SetScale on both halves
Boot *newBootBL(const std::vector<int64_t> &sk, const Seed256 &seed, int device);
// ct_res0/1: level-1 results of the two baseline convolutions, device [2][2][N] over (Q0, Q1 of set [7]) at `scale`; out0/1 likewise at level 1
void blBootReLU(Boot *B, const uint64_t *ct_res0, const uint64_t *ct_res1, double scale, double alpha, double pow, uint64_t *out0, uint64_t *out1, double *out_scale);
// eval.go:272-607 for kinds "Conv", "Conv_sparse", "StrConv_sparse" (hconv_resnet.cpp); returns a level-1, scale-2^30 ciphertext
// evalConv_BN / evalConv_BNRelu_new for the images of a batch: kernel plaintexts prepared once, hc_conv_then_pack_batch, the tail as one launch set
std::vector<Ciphertext> evalConv_BN_batch(Context *cont, const std::vector<Ciphertext> &ct_inputs, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                                          const std::vector<double> &bn_b, int in_wid, int ker_wid, int real_ib, int real_ob, int norm, double out_scale, bool trans);
std::vector<Ciphertext> evalConv_BNRelu_new_batch(Context *cont, const std::vector<Ciphertext> &ct_inputs, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                                                  const std::vector<double> &bn_b, double alpha, double pow, int in_wid, int kp_wid, int ker_wid, int real_ib, int real_ob,
                                                  int norm, int log_sparse, const std::string &kind);
int imageBatch();                                // HCONV_IMAGE_BATCH (1..8; default 1): images that go through a layer as one launch set
Ciphertext evalConv_BNRelu_new(Context *cont, const Ciphertext &ct_input, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                               const std::vector<double> &bn_b, double alpha, double pow, int in_wid, int kp_wid, int ker_wid, int real_ib, int real_ob,
                               int norm, int log_sparse, const std::string &kind);
// test.go:76-370 — `resnet ker depth 1 n false`
void testResNet_crop_sparse(int st, int end, int ker_wid, int depth, bool debug);
// test_BL.go:16 — the slot-packed baseline the reference runs first (hconv_bl.cpp); boot = true adds Bootstrapp + ReLU (test_BL.go:113-168)
void testConv_BL_in(int real_batch, int in_wid, int ker_wid, int total_test_num, bool boot);

// ckks.Evaluator subset used by conv.go, one C-ABI call per limb row
class GpuEvaluator {
public:
    explicit GpuEvaluator(Context *c) : cont(c) {}
    Ciphertext MulNew(const Ciphertext &ct, const Plaintext &pt);            // conv.go:527, 288
    void SetScale(Ciphertext &ct, double scale);                             // conv.go:528
    Ciphertext SubNew(const Ciphertext &a, const Ciphertext &b);             // conv.go:289
    void Add(const Ciphertext &a, const Ciphertext &b, Ciphertext &out);     // conv.go:290, 292
    void AddPlain(const Ciphertext &a, const Plaintext &b, Ciphertext &out); // eval.go:258
    void RotateGal(const Ciphertext &ct, uint64_t galEl, Ciphertext &out);   // conv.go:291
private:
    Context *cont;
    Ciphertext alloc(int level, double scale);
};

}  // namespace hconv
