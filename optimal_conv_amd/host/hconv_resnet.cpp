// hconv_resnet.cpp — `resnet <ker> <depth> 1 <n> false` (scope row 8f-3): the reference's encrypted ResNet inference
// (test.go:76-370 testResNet_crop_sparse) on the MI355X engine, and the layer operator it is built from
// (eval.go:272-607 evalConv_BNRelu_new for kinds "Conv_sparse" / "StrConv_sparse").
//
// Reference mapping (file:line -> here):
//   main.go:609-621   CLI `resnet ker depth wide_case test_num cf100` (wide_case 1)        -> conv_main.cpp -> testResNet_crop_sparse
//   test.go:76-370    testResNet_crop_sparse: 7+1+5+1+5 conv-BN-ReLU layers (depth 20), reduce-mean + FC as one convolution,
//                     file formats Resnet_weights/.../w{i}-{conv,a,b}.csv, final-fckernel.csv, final-fcbias.csv,
//                     Resnet_plain_data/.../test_image_{i}.csv, Resnet_enc_results/.../class_result_ker{k}_{i}.csv -> same
//   eval.go:335-392   StrConv_sparse front end: two convolutions on the even / odd output channels at norm/2, X^(norm/4)
//                     shift, add, offset monomial                                                -> evalConv_BNRelu_new
//   eval.go:437-565   bootstrapping (sparse slots), evalReLU, keep_ctxt / ext_double_ctxt, StoC   -> hconv_relu.cpp
//   main.go:920-939   prt_mat_one_norm                                                            -> prt_mat_one_norm
// Weights and images are not shipped with the reference (README.md:23); tests/golden/gen_resnet_csv.py writes synthetic
// ones in the same file layout together with the plain float model's class scores.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <thread>

#include "hconv_host.hpp"
#include "hconv_sha256.hpp"

namespace hconv {

#define HCX(c, call) do { int rc_ = (call); if (rc_) panic(std::string(#call) + ": " + hc_last_error(c)); } while (0)
static std::string dur(std::chrono::steady_clock::time_point t0) {
    double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    char b[64];
    if (ns < 1e6) snprintf(b, sizeof b, "%.6gµs", ns / 1e3); else if (ns < 1e9) snprintf(b, sizeof b, "%.9gms", ns / 1e6); else snprintf(b, sizeof b, "%.9gs", ns / 1e9);
    return b;
}
static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }
static double secs(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

// MulNew(ct, EncodeCoeffs(+-X^idx at scale 1)) on a level-0 ciphertext (eval.go:361-367, 374-387), in place
static void mul_monomial_l0(Context *c, Ciphertext &ct, int idx, bool negative) {
    std::vector<uint64_t> m((size_t)N, 0); m[(size_t)idx] = negative ? PARAMS6_Q[0] - 1 : 1;
    void *v = nullptr; HCX(c->hc, hc_malloc(c->hc, (size_t)N * 8, &v)); uint64_t *pt = (uint64_t *)v;
    HCX(c->hc, hc_upload(c->hc, pt, m.data(), (size_t)N * 8)); HCX(c->hc, hc_ntt(c->hc, 0, pt, pt, 1));
    for (int d = 0; d < 2; d++) HCX(c->hc, hc_mul(c->hc, 0, ct.d + (size_t)d * N, pt, ct.d + (size_t)d * N, 1));
    HCX(c->hc, hc_free(c->hc, pt));
}

// eval.go:272-607 for the kinds the conv / convReLU / resnet command lines reach, for the images of a batch (ct_inputs.size() <= HCONV_IMAGE_BATCH): every image
// sees the operations of the reference's per-image flow, but a launch covers all of them (hc_conv_then_pack_batch; hc_set_batch inside the tail)
std::vector<Ciphertext> evalConv_BNRelu_new_batch(Context *cont, const std::vector<Ciphertext> &ct_inputs, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                                                  const std::vector<double> &bn_b, double alpha, double pow_, int in_wid, int kp_wid, int ker_wid, int real_ib, int real_ob,
                                                  int norm, int log_sparse, const std::string &kind) {
    if (!cont->btp) panic("evalConv_BNRelu_new needs a context built with boot = true");
    const int nimg = (int)ct_inputs.size();
    const double out_scale = exp2(round(log2((double)PARAMS6_Q[0]) - (pow_ + 8)));                       // eval.go:433
    std::vector<Ciphertext> ct_conv;
    if (kind == "StrConv_sparse") {                                                                       // eval.go:335-392 (modify_ker, !full)
        std::vector<double> a0((size_t)real_ob / 2), a1((size_t)real_ob / 2), b0((size_t)real_ob / 2), b1((size_t)real_ob / 2);
        for (int i = 0; i < real_ob / 2; i++) { a0[(size_t)i] = bn_a[(size_t)(2 * i)]; a1[(size_t)i] = bn_a[(size_t)(2 * i + 1)]; b0[(size_t)i] = bn_b[(size_t)(2 * i)]; b1[(size_t)i] = bn_b[(size_t)(2 * i + 1)]; }
        std::vector<double> k0(ker_in.size() / 2), k1(ker_in.size() / 2);
        for (int k = 0; k < ker_wid * ker_wid; k++) for (int i = 0; i < real_ib; i++) for (int j = 0; j < real_ob / 2; j++) {
            k0[(size_t)(k * real_ib * real_ob / 2 + (i * real_ob / 2 + j))] = ker_in[(size_t)(k * real_ib * real_ob + (i * real_ob + 2 * j))];
            k1[(size_t)(k * real_ib * real_ob / 2 + (i * real_ob / 2 + j))] = ker_in[(size_t)(k * real_ib * real_ob + (i * real_ob + 2 * j + 1))];
        }
        std::vector<Ciphertext> r1 = evalConv_BN_batch(cont, ct_inputs, k0, a0, b0, in_wid, ker_wid, real_ib, real_ob / 2, norm / 2, out_scale, false);
        std::vector<Ciphertext> r2 = evalConv_BN_batch(cont, ct_inputs, k1, a1, b1, in_wid, ker_wid, real_ib, real_ob / 2, norm / 2, out_scale, false);
        const int max_batch = N / (in_wid * in_wid);
        for (int z = 0; z < nimg; z++) {
            mul_monomial_l0(cont, r2[(size_t)z], norm / 4, false);                                        // eval.go:361-367
            for (int d = 0; d < 2; d++) HCX(cont->hc, hc_add(cont->hc, 0, r1[(size_t)z].d + (size_t)d * N, r2[(size_t)z].d + (size_t)d * N, r1[(size_t)z].d + (size_t)d * N, 1));   // AddNew (369)
            freeCt(cont, r2[(size_t)z]);
            if ((in_wid - ker_wid / 2) % 2 == 0) mul_monomial_l0(cont, r1[(size_t)z], N - max_batch * (in_wid + 1), true);                        // eval.go:377-387
        }
        ct_conv = r1;
    } else if (kind == "Conv_sparse" || kind == "Conv") {
        ct_conv = evalConv_BN_batch(cont, ct_inputs, ker_in, bn_a, bn_b, in_wid, ker_wid, real_ib, real_ob, norm, out_scale, false);             // eval.go:433
    } else panic("No kind!");
    // hand-over to the bootstrapper's context (its own stream): everything queued on the convolution context - the stride layers end on a product and an addition that
    // nothing has waited for - must be complete before the other stream reads ct_conv
    HCX(cont->hc, hc_sync(cont->hc));
    std::vector<const uint64_t *> conv_d; for (const Ciphertext &c : ct_conv) conv_d.push_back(c.d);
    std::vector<BootCiphertext> r = evalConv_BNRelu_tail_batch(cont->btp, kind, log_sparse, conv_d, ct_conv[0].Scale, alpha, pow_, in_wid, kp_wid);
    for (Ciphertext &c : ct_conv) freeCt(cont, c);
    // [2][2][N] over (Q0, Q1): what the next convolution reads. The result blocks belong to the bootstrapper's context and go back to it; the layer's outputs are
    // blocks of the convolution context (a block released into a context that did not allocate it would leave the owner's block table pointing at memory it no longer
    // owns - the lifetime bug hc_free's stream synchronisation used to hide). The tail has synchronised its stream, so the copies see the finished results.
    std::vector<Ciphertext> outs((size_t)nimg);
    for (int z = 0; z < nimg; z++) {
        Ciphertext &out = outs[(size_t)z]; out.level = r[(size_t)z].level; out.Scale = r[(size_t)z].Scale;
        const size_t bytes = (size_t)2 * (out.level + 1) * N * 8;
        { void *v = nullptr; HCX(cont->hc, hc_malloc(cont->hc, bytes, &v)); out.d = (uint64_t *)v; }
        HCX(cont->hc, hc_copy(cont->hc, out.d, r[(size_t)z].d, bytes));
    }
    HCX(cont->hc, hc_sync(cont->hc));
    for (BootCiphertext &b : r) freeBootCt(cont->btp, b);
    return outs;
}
Ciphertext evalConv_BNRelu_new(Context *cont, const Ciphertext &ct_input, const std::vector<double> &ker_in, const std::vector<double> &bn_a,
                               const std::vector<double> &bn_b, double alpha, double pow_, int in_wid, int kp_wid, int ker_wid, int real_ib, int real_ob,
                               int norm, int log_sparse, const std::string &kind) {
    return evalConv_BNRelu_new_batch(cont, {ct_input}, ker_in, bn_a, bn_b, alpha, pow_, in_wid, kp_wid, ker_wid, real_ib, real_ob, norm, log_sparse, kind)[0];
}

// main.go:920-939
static std::vector<double> prt_mat_one_norm(const std::vector<double> &vec, int batch, int norm, int sj, int sk) {
    const int mat_size = (int)vec.size() / batch; std::vector<double> out;
    int j = 1, k = 1;
    for (size_t i = 0; i < vec.size(); i += (size_t)batch) {
        if (j == sj && k == sk) {
            out.assign((size_t)(batch / norm), 0.0);
            for (size_t idx = 0; idx < out.size(); idx++) out[idx] = vec[i + (size_t)norm * idx];
            for (double v : out) printf("%.10f ", v);
            printf("\n");
        }
        k++;
        if (k * k > mat_size) { k = 1; j++; }
    }
    return out;
}
// main.go:992-1005
static void writeTxt(const std::string &name, const std::vector<double> &v) {
    std::ofstream f(name, std::ios::trunc);
    if (!f) { fprintf(stderr, "failed creating file: %s\n", name.c_str()); exit(1); }
    char buf[64];
    for (double d : v) { snprintf(buf, sizeof buf, "%.17g\n", d); f << buf; }
}

// test.go:76-370: `resnet ker depth 1 n false` (BASELINE config 5). The wide drivers (wide_case 2 / 3: testResNet_crop_sparse_wide, test.go:638-912) and the CIFAR-100 head
// (cf100: two final convolutions, test.go:287-315) are SURVEY section 2 row 14 - out of scope - and were removed from this host in round 6 (they lived here in rounds 1-5).
void testResNet_crop_sparse(int st, int end, int ker_wid, int depth, bool debug) {
    if (const char *fi = testOnlyEnv("HCONV_RESNET_FIRST_IMAGE")) st = atoi(fi);      // test mode: images st .. end - 1 (one image of a batch run alone, for the batch == single digest test)
    (void)debug;
    const std::string ker_name = "ker" + std::to_string(ker_wid), tag = "crop_" + ker_name + "_d" + std::to_string(depth) + "_wid1/";
    const std::string weight_dir = "Resnet_weights/weights_" + tag, out_dir = "Resnet_enc_results/results_" + tag, img_dir = "Resnet_plain_data/" + tag;
    const int fc_out = 10; const double init_pow = 6.0, mid_pow = 6.0, final_pow = 6.0;                  // test.go:84-89 (cifar10)
    int num_blcs[3];
    if (depth == 20) { num_blcs[0] = 7; num_blcs[1] = 5; num_blcs[2] = 5; } else if (depth == 14) { num_blcs[0] = 5; num_blcs[1] = 3; num_blcs[2] = 3; }
    else if (depth == 8) { num_blcs[0] = 3; num_blcs[1] = 1; num_blcs[2] = 1; } else panic("wrong depth (not in 8, 14, 20)!");
    const int real_batch[3] = {16, 32, 64}, norm[3] = {4, 8, 16}, log_sparse[3] = {2, 3, 4};           // test.go:107-112
    const int logN = 16; const double alpha = 0.0;
    const std::vector<int> in_wids = {32, 16, 8}, raw_in_wids = {32 - ker_wid / 2, 16 - ker_wid / 2, 8 - ker_wid / 2};
    const int ker_size = ker_wid * ker_wid;
    int max_batch[3]; for (int i = 0; i < 3; i++) max_batch[i] = (1 << logN) / (in_wids[(size_t)i] * in_wids[(size_t)i]);
    mkdir("Resnet_enc_results", 0755); mkdir(out_dir.c_str(), 0755);
    auto W = [&](int i, const char *what, int size) { return readTxt(weight_dir + "w" + std::to_string(i) + "-" + what + ".csv", size); };
    const char *kind_name = "Resnet_crop_sparse";

    // HCONV_IMAGE_THREADS=K (not a reference feature): K host threads, each with its own context (keys, bootstrappers, stream),
    // classify disjoint shares of the images at the same time. A layer's launches are mostly far below one wave of workgroups, so the
    // images of several streams overlap on the CUs (separate PROCESSES time-slice the device instead: tools/resnet_throughput.py).
    // Measured on MI355X, ResNet-20, 24 images, HCONV_ASYNC_ALLOC=1: 1 thread 4 704 images/hour, 2 threads 5 982, 3 threads 5 553,
    // 4 threads 5 340 (each image is ~60 000 kernel launches; the threads share the runtime's launch path).
    const int nbatch = imageBatch();
    const int n_threads = std::max(1, std::min(getenv("HCONV_IMAGE_THREADS") ? atoi(getenv("HCONV_IMAGE_THREADS")) : 1, (end - st + nbatch - 1) / nbatch));
    if (n_threads > 1) setenv("HCONV_ASYNC_ALLOC", "1", 0);      // several contexts in one process: cached allocations on non-blocking streams, or every hipFree stalls all of them
    std::mutex mu; std::condition_variable cv; int ready = 0; std::chrono::steady_clock::time_point t_go;
    auto run_images = [&](int tix) {
    Context *cont;
    { std::lock_guard<std::mutex> g(mu); cont = newContext(logN, ker_wid, in_wids, raw_in_wids, true, kind_name); }      // one at a time: key generation prints and allocates
    { std::unique_lock<std::mutex> g(mu); if (++ready == n_threads) { t_go = now(); cv.notify_all(); } else cv.wait(g, [&] { return ready == n_threads; }); }
    // images go through the network in groups of HCONV_IMAGE_BATCH (every layer's launches cover the whole group); thread tix takes groups tix, tix + n_threads, ...
    for (int g0 = st + tix * nbatch; g0 < end; g0 += n_threads * nbatch) {
        const int nimg = std::min(nbatch, end - g0);
        std::vector<Ciphertext> ct_layer;
        auto enc_start = now();
        for (int iter = g0; iter < g0 + nimg; iter++) {
            printf("Running  %d -th iter... ker size:  %d\n", iter, ker_wid);
            std::vector<double> image = readTxt(img_dir + "test_image_" + std::to_string(iter) + ".csv", in_wids[0] * in_wids[0] * 3);
            std::vector<double> input((size_t)N, 0.0); int k = 0;
            for (int i = 0; i < in_wids[0]; i++) for (int j = 0; j < in_wids[0]; j++) for (int b = 0; b < 3; b++) {
                if (i < raw_in_wids[0] && j < raw_in_wids[0]) input[(size_t)(i * in_wids[0] * max_batch[0] + j * max_batch[0] + b * norm[0])] = image[(size_t)k];   // sparse pack the input
                k++;
            }
            if (resnetReplaySeed()) cont->replay_encryptions = iter;     // test mode: the encryption randomness of an image depends on its index only, not on which images ran before it in this process
            ct_layer.push_back(EncryptNew(cont, EncodeCoeffs(input, cont->ECD_LV, cont->scale), cont->ECD_LV, cont->scale));
        }
        printf("vec size:  %d\n", N); printf("input width:  [%d %d %d]\n", raw_in_wids[0], raw_in_wids[1], raw_in_wids[2]);
        printf("kernel width:  %d\n", ker_wid); printf("num batches:  [%d %d %d]\n", real_batch[0], real_batch[1], real_batch[2]);
        printf("Encryption done in %s \n", dur(enc_start).c_str());
        double timings[6]; auto begin_start = now(), start = now();
        int layer_no = 0;
        auto step = [&](std::vector<Ciphertext> next) {
            for (Ciphertext &c : ct_layer) freeCt(cont, c);
            ct_layer = std::move(next);
            if (resnetReplaySeed()) for (size_t z = 0; z < ct_layer.size(); z++) {      // test mode: SHA-256 of the ciphertext the layer hands on ([2][level+1][N], as tests/parity_cases.py sha_ct)
                std::vector<uint64_t> rows((size_t)2 * (ct_layer[z].level + 1) * N); HCX(cont->hc, hc_download(cont->hc, rows.data(), ct_layer[z].d, rows.size() * 8));
                Sha256 h; h.update(rows.data(), rows.size() * 8);
                printf("replay digest layer %d image %d level %d scale %.17g %s\n", layer_no, g0 + (int)z, ct_layer[z].level, ct_layer[z].Scale, h.hex().c_str());
            }
            layer_no++;
        };

        double pow_ = init_pow;                                                                          // ResNet Block 1
        for (int i = 1; i <= num_blcs[0]; i++) {
            const int ib = i == 1 ? 3 : real_batch[0], ob = real_batch[0];
            step(evalConv_BNRelu_new_batch(cont, ct_layer, W(i - 1, "conv", ib * ob * ker_size), W(i - 1, "a", ob), W(i - 1, "b", ob),
                                           alpha, pow_, in_wids[0], raw_in_wids[0], ker_wid, ib, ob, norm[0], log_sparse[0], "Conv_sparse"));
            pow_ = mid_pow;
            printf("Block1, Layer  %d done!\n", i);
        }
        printf("Block1 done.\n"); timings[0] = secs(start); start = now();
        step(evalConv_BNRelu_new_batch(cont, ct_layer, W(num_blcs[0], "conv", real_batch[0] * real_batch[1] * ker_size), W(num_blcs[0], "a", real_batch[1]), W(num_blcs[0], "b", real_batch[1]),
                                       alpha, pow_, in_wids[0], raw_in_wids[1], ker_wid, real_batch[0], real_batch[1], norm[1], log_sparse[0] - 1, "StrConv_sparse"));           // test.go:200
        printf("Block1 to 2 done!\n"); timings[1] = secs(start); start = now();
        for (int i = 1; i <= num_blcs[1]; i++) {                                                         // ResNet Block 2
            const int w = num_blcs[0] + i;
            step(evalConv_BNRelu_new_batch(cont, ct_layer, W(w, "conv", real_batch[1] * real_batch[1] * ker_size), W(w, "a", real_batch[1]), W(w, "b", real_batch[1]),
                                           alpha, pow_, in_wids[1], raw_in_wids[1], ker_wid, real_batch[1], real_batch[1], norm[1], log_sparse[1], "Conv_sparse"));
            printf("Block2, Layer  %d done!\n", i);
        }
        printf("Block2 done.\n"); timings[2] = secs(start); start = now();
        { const int w = num_blcs[0] + num_blcs[1] + 1;
          step(evalConv_BNRelu_new_batch(cont, ct_layer, W(w, "conv", real_batch[1] * real_batch[2] * ker_size), W(w, "a", real_batch[2]), W(w, "b", real_batch[2]),
                                         alpha, pow_, in_wids[1], raw_in_wids[2], ker_wid, real_batch[1], real_batch[2], norm[2], log_sparse[1] - 1, "StrConv_sparse")); }               // test.go:225
        printf("Block2 to 3 done!\n"); timings[3] = secs(start); start = now();
        for (int i = 1; i <= num_blcs[2]; i++) {                                                         // ResNet Block 3
            const int w = num_blcs[0] + num_blcs[1] + i + 1;
            if (i == num_blcs[2]) pow_ = final_pow;
            step(evalConv_BNRelu_new_batch(cont, ct_layer, W(w, "conv", real_batch[2] * real_batch[2] * ker_size), W(w, "a", real_batch[2]), W(w, "b", real_batch[2]),
                                           alpha, pow_, in_wids[2], raw_in_wids[2], ker_wid, real_batch[2], real_batch[2], norm[2], log_sparse[2], "Conv_sparse"));
            printf("Block3, Layer  %d done!\n", i);
        }
        printf("Block3 done.\n"); timings[4] = secs(start); start = now();

        int ker_inf_wid = raw_in_wids[2]; if (ker_inf_wid % 2 == 0) ker_inf_wid++;                       // test.go:279-334: reduce_mean + FC
        std::vector<double> ker_inf = readTxt(weight_dir + "final-fckernel.csv", real_batch[2] * fc_out);
        std::vector<double> bn_bf = readTxt(weight_dir + "final-fcbias.csv", fc_out);
        std::vector<Ciphertext> ct_result;
        {                                                                                                // test.go:316-334
            std::vector<double> ker_inf_((size_t)(ker_inf_wid * ker_inf_wid * real_batch[2] * fc_out));
            for (size_t i = 0; i < ker_inf.size(); i++) for (int b = 0; b < ker_inf_wid * ker_inf_wid; b++) ker_inf_[i + (size_t)b * real_batch[2] * fc_out] = ker_inf[i];
            std::vector<double> bn_af((size_t)fc_out, 1.0 / (double)(raw_in_wids[2] * raw_in_wids[2]));
            ct_result = evalConv_BN_batch(cont, ct_layer, ker_inf_, bn_af, bn_bf, in_wids[2], ker_inf_wid, real_batch[2], fc_out, norm[2], (double)(1 << 30), false);
        }
        printf("Final FC done.\n"); timings[5] = secs(start); start = now();
        printf("\n===============  DECRYPTION  ===============\n\n");
        for (int z = 0; z < nimg; z++) {
            std::vector<double> res_tmp = DecryptDecodeCoeffs(cont, ct_result[(size_t)z]);
            printf("Decryption Done in %s \n", dur(start).c_str());
            std::vector<double> res_out = prt_mat_one_norm(res_tmp, max_batch[2], norm[2], ker_inf_wid / 2 + 1, ker_inf_wid / 2 + 1);
            res_out.resize((size_t)fc_out);
            printf("\n result:  ["); for (double v : res_out) printf("%.10f ", v);
            printf("]\n");
            writeTxt(out_dir + "class_result_" + ker_name + "_" + std::to_string(g0 + z) + ".csv", res_out);
            freeCt(cont, ct_layer[(size_t)z]); freeCt(cont, ct_result[(size_t)z]);
        }
        printf("Blc1:  %g  sec\nBlc1->2:  %g  sec\nBlc2:  %g  sec\nBlc2->3:  %g  sec\nBlc3:  %g  sec\nFinal (reduce_mean & FC):  %g  sec\n", timings[0], timings[1], timings[2], timings[3], timings[4], timings[5]);
        printf("Total done in %s %s\n", dur(begin_start).c_str(), nimg > 1 ? ("(" + std::to_string(nimg) + " images)").c_str() : "");
    }
    { std::lock_guard<std::mutex> g(mu); freeContext(cont); }
    };
    if (n_threads == 1) { run_images(0); return; }
    std::vector<std::thread> th; for (int t = 0; t < n_threads; t++) th.emplace_back(run_images, t);
    for (auto &t : th) t.join();
    // all contexts were ready at t_go; context release is included in the figure below (a few hipFree), image work dominates
    printf("All %d images done in %s  (%d image threads)\n", end - st, dur(t_go).c_str(), n_threads);
}

}  // namespace hconv
