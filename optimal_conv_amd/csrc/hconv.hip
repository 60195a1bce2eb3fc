// hconv.hip — C ABI (include/hconv.h) + host orchestration of the gfx950 kernels in hc_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o optimal_conv_amd/libhconv.so csrc/hconv.hip
// There is deliberately no CPU path in this library: every entry point that computes launches HIP kernels.
#ifdef HC_EMU
#include "hip_emu.h"   // tests/kernel_emu: CPU fiber stand-in used only by -m "not gpu" tests
#else
#include <hip/hip_runtime.h>
#endif

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <utility>
#include <string>
#include <vector>

#include "../../include/hconv.h"
#include "../../include/hconv_test_hooks.h"
#include "hc_kernels.h"
#include "hc_gomath.h"
// workgroups per row of the streaming (grid-stride) kernels of the leveled evaluator and the key switch: a thread strides over 65536 / (256 x this) coefficients.
// Measured on the convReLU 5 1 tail (profiles/round4_chain_occupancy_ab.txt, rounds 7-8): the inner products and the giant-step sums at 256 (one coefficient per thread)
// and the rotation finishes at 128 take 2.3 % off a layer against 64 / 32; per kernel (HIP-event profile): the extension's source side (basis_yv, mdrs) and the tensor product gain at 256, the polynomial leaves at
// 128, the plain pointwise operations at 32 (at 256 they take twice as long: a workgroup's fixed cost against 256 coefficients of work)
#ifndef HC_GX_LV
#define HC_GX_LV 64
#endif
#ifndef HC_GX_PW
#define HC_GX_PW 32
#endif
#ifndef HC_GX_YV
#define HC_GX_YV 256
#endif
#ifndef HC_GX_TEN
#define HC_GX_TEN 256
#endif
#ifndef HC_GX_LIN
#define HC_GX_LIN 128
#endif
#ifndef HC_GX_ROT
#define HC_GX_ROT 128
#endif
#ifndef HC_GX_QPMS
#define HC_GX_QPMS 256
#endif
#ifndef HC_GX_MAC
#define HC_GX_MAC 256
#endif
#ifndef HC_GX_MACM
#define HC_GX_MACM 256
#endif

#define HC_N 65536
#define HC_LOGN 16

// ------------------------------------------------------------------ host number theory (table construction)
static u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 h_powmod(u64 b, u64 e, u64 q) {
    u64 r = 1; b %= q;
    while (e) { if (e & 1) r = h_mulmod(r, b, q); b = h_mulmod(b, b, q); e >>= 1; }
    return r;
}
static u64 h_inv(u64 a, u64 q) { return h_powmod(a % q, q - 2, q); }
static u64 h_shoup(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }
static HcTw h_pair(u64 w, u64 q) { HcTw p; p.w = w; p.ws = h_shoup(w, q); return p; }
static bool h_is_prime(u64 n) {
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n % p == 0) return n == p; }
    u64 d = n - 1; int s = 0; while ((d & 1) == 0) { d >>= 1; s++; }
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = h_powmod(a, d, n); if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s; i++) { x = h_mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}
// Same rule as the reference's dependency (ring.primitiveRoot): candidates 3,4,5,... and the
// first g with g^((q-1)/f) != 1 for every prime factor f of q-1 wins. psi = g^((q-1)/2N).
static u64 h_primitive_root(u64 q) {
    std::vector<u64> fac; u64 n = q - 1;
    for (u64 p = 2; p * p <= n; p += (p == 2 ? 1 : 2)) if (n % p == 0) { fac.push_back(p); while (n % p == 0) n /= p; }
    if (n > 1) fac.push_back(n);
    for (u64 g = 3;; g++) {
        bool ok = true;
        for (u64 f : fac) if (h_powmod(g, (q - 1) / f, q) == 1) { ok = false; break; }
        if (ok) return g;
    }
}
static u32 h_bitrev16(u32 x) { u32 r = 0; for (int i = 0; i < 16; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// ------------------------------------------------------------------ context
struct HcModHost {
    HcMod m;
    u64 psi, psi_inv;
    HcTwTab fwd, inv;            // device tables
    HcTwTab32 fwd32 = {nullptr, nullptr, nullptr, nullptr}, inv32 = {nullptr, nullptr, nullptr, nullptr};     // moduli below 2^31: the same tables as 8-byte entries (HC_S32)
    HcTwTab inv_f64;             // moduli below 2^49: the inverse tables as {w, w/q} doubles (fp64 inverse transform of loop A)
    std::vector<void *> allocs;
};
struct HcEvk { HcTw *q_rows; HcTw *p_rows; bool row_local; bool row256; };   // row_local: the permutation stays inside 4096-coefficient tiles; row256: even inside 256-coefficient rows   // [2][N] each, Shoup pairs: q_rows (key / P mod Q0) natural order; p_rows (key / N mod P) lo-local order
struct HcSwk { u64 *rows = nullptr; int level = 0, beta = 0; };   // general switching key: [beta][2][level+1+np][N], stored form
struct HcProfRec { std::string name; hipEvent_t a, b; };
struct hc_ctx {
    int device = 0, nq = 0, np = 0;
    hipStream_t stream = nullptr;
    std::vector<HcModHost> mods;
    std::map<u64, HcEvk> evk;
    std::map<u64, HcSwk> swk;
    HcTw *idx_pairs = nullptr;   // [logN][N] idx plaintexts as Shoup pairs
    // workspace
    u64 *ws_cts = nullptr; size_t ws_cts_rows = 0;     // loop A output / tree ping
    u64 *ws_cts2 = nullptr; size_t ws_cts2_rows = 0;   // tree pong
    HcTw *ws_ctc = nullptr; size_t ws_ctc_cts = 0;     // [batch][2 polys][2 limbs][N] ct_in times the MultByConst constants, Shoup pairs
    u64 *ws_tmp = nullptr; size_t ws_tmp_rows = 0;
    HcMod *d_mods = nullptr; HcTw *d_csts = nullptr;      // device copies: all moduli (Q then P); per-call constants of the leveled ops
    struct CacheBlk { size_t n = 0; hipEvent_t ev = nullptr; bool pending = false; };
    std::map<char *, CacheBlk> cache_blk; std::map<size_t, std::vector<void *>> cache_free;      // HCONV_ASYNC_ALLOC=1: sizes of the blocks this context allocated; parked blocks by size
    int async_alloc = 0;                                    // option async_alloc = 1: non-blocking stream + cached allocations (see hcx_malloc)
#ifdef HC_EMU
    long small_mm_wgs = 0;                                  // the CPU emulator pays per fiber switch, and the quarter-tile rows passes exchange through 24 of them: the emulated suites run the 16-row
                                                            // kernels unless a test asks (tests/test_emu_parity.py::test_batched_transforms_on_quarter_tiles_or_not forces each form)
#else
    long small_mm_wgs = 1024;                               // option small_mm_wgs: a batched inverse pass / second forward pass of at most this many 16-row workgroups runs on quarter tiles (0: never); measured 512 .. 32 768: profiles/round6_chain_probes.txt
#endif
    long allocs_live = 0;                                   // hc_malloc blocks not yet freed (the allocation mode may only change while there are none)
    HcRowMod *d_rowmods = nullptr;                          // per modulus: both twiddle tables + q, mu (multi-modulus batched transforms)
    u64 *ws_mm = nullptr; size_t ws_mm_rows = 0;            // scratch of the batched key switch / rescale
    u64 *ws_accm = nullptr; size_t ws_accm_rows = 0;        // inner products of several hoisted rotations (hc_keyswitch_qp_rotate_many)
    struct KsPlan { HcBasisExt *bx = nullptr, *bxdown = nullptr; HcTw *pinv = nullptr, *pmod = nullptr, *pinv_qlinv = nullptr, *yinv = nullptr, *yinv1 = nullptr; };    // yinv[l]: (S/q_l)^-1 mod q_l, S = the product of the limbs of l's digit (the source side of the extension: hc_k_cols_inv_canon_mm's scale); entries nl..nl+alpha-1: the same for the P limbs (ModDown)     // pmod: P mod q_i; pinv_qlinv: (P q_level)^-1 mod q_i, i < level
    std::map<int, KsPlan> ks_plan;                          // per level: basis-extension constants of every (digit, target limb)
    std::map<int, HcTw *> rescale_plan;                     // per level: qL^-1 mod q_i
    int nb = 1; size_t bs_poly = 0, bs_qp = 0;              // image batch of the leveled entry points (hc_set_batch): images, words between the images of a polynomial / of an extended-basis pair
    const void *hoist_cx = nullptr; int hoist_level = -1;   // the polynomial whose digit decomposition ws_mm currently holds
    long chunk_nodes = 64;
    hipEvent_t ev_fork = nullptr;
    HcCplx *enc_roots = nullptr; int *enc_rot_group = nullptr;      // slot encoder tables (hc_encode_slots), built at first use
    hipEvent_t ev_shard = nullptr;     // hc_conv_then_pack_sharded: this device's partial ciphertext is complete / has been collected
    u64 *ws_gather = nullptr; size_t ws_gather_rows = 0;
    long small_levels = 16;               // pack-tree launches of at most this many nodes (summed over the batch) run on the 1024-thread S kernels; 0 = never
    long peer_access = 1;                 // hc_conv_then_pack_sharded: enable direct peer copies between distinct devices (0: leave the copies to hipMemcpyPeerAsync's staging)
    int xcd_rows = 1;                     // XCD-aware 1-D grid of the rows passes (HcMm::xcd; 0 = the plain 3-D grid, kept for A/B builds)
    int pack32 = 1;                       // 1: library-internal rows of moduli below 2^31 as 4-byte words (hc_kernels.h hc_ld32): transform seams, extended digits, switching keys. 2: the rows of the caller's leveled
                                          // operands as well (include/hconv.h "4-byte rows"; option pack32). 0: off (A/B)
    int rot_fuse = 1;                     // hc_keyswitch_qp_rotate_many: the rotations' tails (+ P c0, permutation) in the inner product's stores (HcRotFin) instead of one hc_k_qp_rotate_finish per rotation (option rot_fuse / HCONV_ROT_FUSE=0 for A/B)
    int small32 = 1;                      // the batched transform kernels take their 32-bit form for rows of a modulus below 2^31 (hc_kernels.h HC_S32; option small32 / HCONV_SMALL32=0 for A/B)
    unsigned peer_warned = 0;             // bit d: enabling peer access to device d failed and was reported once
    unsigned peer_enabled = 0;            // bit d: peer access from this context's device to device d was enabled by (or found enabled for) this context
    long profile = 0;
    std::vector<HcProfRec> prof;
    std::map<std::string, std::pair<double, long>> prof_acc;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    std::string err;
};
struct hc_ker { u64 *d = nullptr; int max_ob = 0; };

static std::string g_create_err;
static int hc_fail(hc_ctx *c, int code, const char *fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}
#define HC_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hc_fail(c, HC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define HC_ENTER(c) do { if (!(c)) return HC_ERR_ARG; HC_HIP(c, hipSetDevice((c)->device)); } while (0)

// forward lazy-reduction mode by modulus size (see HC_FM_* in hc_kernels.h): FREE needs 74q < 2^64 <=> q < 2^57 (hc_fm_free); ALT needs 8q < 2^64
// Allocation: plain hipMalloc / hipFree by default. hipFree synchronises the whole device, which is harmless with one context but
// serialises independent contexts driven from several host threads (a thread's free waits for every other thread's queued work).
// Option async_alloc = 1 (set right after hc_ctx_create) gives this context a non-blocking stream and a cache of its own hipMalloc blocks: hc_free
// parks a block, the next request of the same size takes it — no hipFree, hence no device-wide synchronisation, in steady state
// (a layer asks for the same sizes again and again). Reuse is safe because every use of a block is queued on this context's stream
// (see hcx_h2d_async for the one host-side exception). ROCm 7.2's own stream-ordered allocator (hipMallocAsync / hipFreeAsync) was
// tried first and is NOT used: with it `conv 3 3` returned wrong loop-A outputs for channels 19..243 as soon as the 256 MiB staging
// block of hc_prep_ker was recycled, with synchronous or asynchronous frees alike, while this cache — the same reuse pattern on plain
// hipMalloc blocks — is bit-exact.
static hipError_t hcx_malloc(hc_ctx *c, void **p, size_t n) {
    if (!(c && c->async_alloc)) return hipMalloc(p, n);
#ifndef HC_EMU
    if (c->async_alloc == 2) return hipMallocAsync(p, n, c->stream);          // diagnostic mode: ROCm's stream-ordered allocator (see tools/repro_mallocasync.hip)
#endif
    auto it = c->cache_free.find(n);
    if (it != c->cache_free.end() && !it->second.empty()) { *p = it->second.back(); it->second.pop_back(); return hipSuccess; }
    hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess) { hc_ctx::CacheBlk b; b.n = n; c->cache_blk[(char *)*p] = b; }
    return e;
}
static hipError_t hcx_free(hc_ctx *c, void *p) {
    if (!p) return hipSuccess;
    if (!(c && c->async_alloc)) return hipFree(p);
#ifndef HC_EMU
    if (c->async_alloc == 2) return hipFreeAsync(p, c->stream);
#endif
    auto it = c->cache_blk.find((char *)p);
    if (it == c->cache_blk.end()) return hipErrorInvalidDevicePointer;      // not a block of this context: its owner's table would keep pointing at it
    // everything queued so far may still read or write the block: remember that point of the stream (see hcx_h2d_async)
    if (!it->second.ev && hipEventCreateWithFlags(&it->second.ev, hipEventDisableTiming) != hipSuccess) it->second.ev = nullptr;
    if (it->second.ev && hipEventRecord(it->second.ev, c->stream) == hipSuccess) it->second.pending = true;
    else { hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) return e; it->second.pending = false; }
    c->cache_free[it->second.n].push_back(p);
    return hipSuccess;
}
// Host-to-device copy "on the stream" from pageable memory. The runtime may carry such a copy out at once from the host instead of
// queueing it. With hipFree that is invisible (it drains the device first); with cached blocks a block parked while kernels that
// use it are still queued could be handed out again and be overwritten by an eager copy before those kernels ran. So a copy into a
// recycled block first waits for the event recorded when the block was parked — normally long complete, so nothing stalls, and
// never anything of another context.
static hipError_t hcx_h2d_async(hc_ctx *c, void *dst, const void *src, size_t n) {
    if (c->async_alloc && !c->cache_blk.empty()) {
        auto it = c->cache_blk.upper_bound((char *)dst);
        if (it != c->cache_blk.begin()) { --it; if ((char *)dst < it->first + it->second.n && it->second.pending) {
            hipError_t e = hipEventSynchronize(it->second.ev); if (e != hipSuccess) return e; it->second.pending = false; } }
    }
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->stream);
}
static hipError_t hcx_h2d(hc_ctx *c, void *dst, const void *src, size_t n) {      // blocking copy on the context's stream (not the null stream)
    hipError_t e = hcx_h2d_async(c, dst, src, n);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}
// Device temporaries of one call: every block allocated through it is released when the call returns, on every path (an early
// HC_HIP / HC_TRY return included); keep() hands a block over to the caller (the result the call produces).
struct HcScratch {
    hc_ctx *c; std::vector<void *> blocks;
    explicit HcScratch(hc_ctx *c_) : c(c_) {}
    ~HcScratch() { if (blocks.empty()) return; hipStreamSynchronize(c->stream); for (void *p : blocks) hcx_free(c, p); }
    template <class T> hipError_t alloc(T **p, size_t bytes) { *p = nullptr; hipError_t e = hcx_malloc(c, (void **)p, bytes); if (e == hipSuccess) blocks.push_back(*p); return e; }
    void keep(void *p) { for (auto it = blocks.begin(); it != blocks.end(); ++it) if (*it == p) { blocks.erase(it); return; } }
};
static inline bool hc_fm_free(u64 q) { return q < (1ull << 57); }     // 74q < 2^64 (hc_ct_round)
static inline bool hc_f64_ok(u64 q) { return q < (1ull << 49); }    // fp64 inverse transform (hc_arith.h): 4q < 2^51

template <int TPB = HC_TPB, class K, class... Args>
static int hc_launch(hc_ctx *c, const char *name, K kernel, dim3 grid, Args... args) {
    hipEvent_t a = nullptr, b = nullptr;
    if (c->profile) { HC_HIP(c, hipEventCreate(&a)); HC_HIP(c, hipEventCreate(&b)); HC_HIP(c, hipEventRecord(a, c->stream)); }
    hipLaunchKernelGGL(kernel, grid, dim3(TPB), 0, c->stream, args...);
    HC_HIP(c, hipGetLastError());
    if (c->profile) { HC_HIP(c, hipEventRecord(b, c->stream)); c->prof.push_back({name, a, b}); }
    return HC_OK;
}
#define HC_TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)
// device-to-device copy on the context's stream
static int hc_copy_d2d(hc_ctx *c, void *dst, const void *src, size_t bytes) {
    HC_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return HC_OK;
}
// pick the kernel instantiation whose forward transform uses the lazy-reduction mode the modulus q allows
#define HC_LAUNCH_FM(q, c, name, KERNEL, grid, ...) \
    (hc_fm_free(q) ? hc_launch(c, name, KERNEL<HC_FM_FREE>, grid, __VA_ARGS__) : hc_launch(c, name, KERNEL<HC_FM_ALT>, grid, __VA_ARGS__))

static int hc_prof_flush(hc_ctx *c) {
    if (c->prof.empty()) return HC_OK;
    HC_HIP(c, hipStreamSynchronize(c->stream));
    for (auto &r : c->prof) {
        float ms = 0; HC_HIP(c, hipEventElapsedTime(&ms, r.a, r.b));
        auto &acc = c->prof_acc[r.name]; acc.first += ms; acc.second += 1;
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    c->prof.clear();
    return HC_OK;
}

template <class T>
static int hc_dev_upload(hc_ctx *c, HcModHost *owner, const std::vector<T> &v, const T **out) {
    void *d = nullptr;
    HC_HIP(c, hcx_malloc(c, &d, v.size() * sizeof(T)));
    HC_HIP(c, hcx_h2d(c, d, v.data(), v.size() * sizeof(T)));
    owner->allocs.push_back(d);
    *out = (const T *)d;
    return HC_OK;
}

// Twiddle tables (layouts documented at HcTwTab). pw[idx] = psi^{bitrev16(idx)} as Lattigo's NttPsi (without
// the Montgomery factor).
static int hc_build_tables(hc_ctx *c, HcModHost *mh, bool inverse) {
    const u64 q = mh->m.q, base = inverse ? mh->psi_inv : mh->psi;
    std::vector<u64> pw(HC_N);
    {
        std::vector<u64> nat(HC_N); nat[0] = 1;
        for (int j = 1; j < HC_N; j++) nat[j] = h_mulmod(nat[j - 1], base, q);
        for (int j = 0; j < HC_N; j++) pw[h_bitrev16((u32)j)] = nat[j];
    }
    std::vector<HcTw> rowsA(256 * 16), rowsB((size_t)256 * 16 * 16), colsA(16), colsB(16 * 16);
    HcTw zero; zero.w = 0; zero.ws = 0;
    for (auto &x : rowsA) x = zero;
    for (auto &x : rowsB) x = zero;
    for (auto &x : colsA) x = zero;
    for (auto &x : colsB) x = zero;
    for (int s = 0; s < 4; s++) for (int g = 0; g < (1 << s); g++) {
        const int slot = (1 << s) - 1 + g;
        colsA[slot] = h_pair(pw[(1 << s) + g], q);
        for (int tid = 0; tid < 16; tid++) colsB[slot * 16 + tid] = h_pair(pw[(16 << s) + tid * (1 << s) + g], q);
        for (int row = 0; row < 256; row++) {
            rowsA[row * 16 + slot] = h_pair(pw[(size_t)(1 << s) * (256 + row) + g], q);
            for (int tid = 0; tid < 16; tid++)
                rowsB[((size_t)row * 16 + slot) * 16 + tid] = h_pair(pw[(size_t)(16 << s) * (256 + row) + tid * (1 << s) + g], q);
        }
    }
    HcTwTab T; memset(&T, 0, sizeof T);
    HC_TRY(hc_dev_upload(c, mh, rowsA, &T.rowsA)); HC_TRY(hc_dev_upload(c, mh, rowsB, &T.rowsB));
    HC_TRY(hc_dev_upload(c, mh, colsA, &T.colsA)); HC_TRY(hc_dev_upload(c, mh, colsB, &T.colsB));
    T.ninv = h_pair(mh->m.ninv, q);
    T.w_last_ninv = h_pair(h_mulmod(pw[1], mh->m.ninv, q), q);
    if (inverse) mh->inv = T; else mh->fwd = T;
    if (q < (1ull << 31)) {             // the 8-byte form for the 32-bit transform bodies: floor(floor(w 2^64 / q) / 2^32) = floor(w 2^32 / q)
        auto narrow = [](const std::vector<HcTw> &v) { std::vector<HcTw32> o(v.size()); for (size_t i = 0; i < v.size(); i++) { o[i].w = (u32)v[i].w; o[i].ws = (u32)(v[i].ws >> 32); } return o; };
        HcTwTab32 S;
        HC_TRY(hc_dev_upload(c, mh, narrow(rowsA), &S.rowsA)); HC_TRY(hc_dev_upload(c, mh, narrow(rowsB), &S.rowsB));
        HC_TRY(hc_dev_upload(c, mh, narrow(colsA), &S.colsA)); HC_TRY(hc_dev_upload(c, mh, narrow(colsB), &S.colsB));
        if (inverse) mh->inv32 = S; else mh->fwd32 = S;
    }
    if (inverse && hc_f64_ok(q)) {      // the same table as doubles: {w, w/q} (both exact inputs, one correctly rounded division)
        auto conv = [&](HcTw x) { HcTw y; y.w = hc_d2u((double)x.w); y.ws = hc_d2u((double)x.w / (double)q); return y; };
        for (auto *v : {&rowsA, &rowsB, &colsA, &colsB}) for (auto &x : *v) x = conv(x);
        HcTwTab U; memset(&U, 0, sizeof U);
        HC_TRY(hc_dev_upload(c, mh, rowsA, &U.rowsA)); HC_TRY(hc_dev_upload(c, mh, rowsB, &U.rowsB));
        HC_TRY(hc_dev_upload(c, mh, colsA, &U.colsA)); HC_TRY(hc_dev_upload(c, mh, colsB, &U.colsB));
        U.ninv = conv(T.ninv); U.w_last_ninv = conv(T.w_last_ninv);
        mh->inv_f64 = U;
    }
    return HC_OK;
}

extern "C" int hc_version(void) { return 2; }      // 2: a plaintext shared by the images of a batch is said by the operation (HC_LV_MUL_PLAIN), never inferred from b0 == b1
extern "C" const char *hc_last_error(const hc_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// the per-modulus table of the batched transforms (HcRowMod); again after option small32 changes
static int hc_upload_rowmods(hc_ctx *c) {
    std::vector<HcRowMod> hr;
    for (auto &mh : c->mods) { HcRowMod r; r.fwd = mh.fwd; r.inv = mh.inv; r.fwd32 = mh.fwd32; r.inv32 = mh.inv32; r.q = mh.m.q; r.mu = mh.m.mu; r.s32 = (c->small32 && mh.m.q < (1ull << 31)) ? 1 : 0; hr.push_back(r); }
    return hcx_h2d(c, c->d_rowmods, hr.data(), hr.size() * sizeof(HcRowMod)) == hipSuccess ? HC_OK : HC_ERR_HIP;
}
extern "C" int hc_ctx_create(hc_ctx **out, int logN, const uint64_t *q, int nq, const uint64_t *p, int np, int device) {
    if (!out || !q || nq < 1 || np < 0 || (np > 0 && !p)) return hc_fail(nullptr, HC_ERR_ARG, "hc_ctx_create: bad arguments");
    // the basis extension's operand registers are sized by the special primes (hc_k_cols_fwd_mm<EXT, NS>, NS = 2 or 5): an NS = 8 build spills 208 bytes per lane and ran
    // 30 % slower - refused rather than shipped as a silently slow path (the reference's parameter sets use 1, 2 or 5 special primes: SURVEY 8(a)-P)
    if (np > HC_MAX_NP) return hc_fail(nullptr, HC_ERR_UNSUPPORTED, "hc_ctx_create: np=%d special primes; this build supports at most %d (the reference's parameter sets use 1, 2 or 5)", np, HC_MAX_NP);
    if (logN != HC_LOGN) return hc_fail(nullptr, HC_ERR_UNSUPPORTED, "hc_ctx_create: logN=%d (this build is specialised for logN=16, the only ring degree the reference CLI uses: main.go:578-579)", logN);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0)
        return hc_fail(nullptr, HC_ERR_HIP, "hc_ctx_create: HIP device %d not available (%d devices) - this library has no CPU path", device, ndev);
    hc_ctx *c = new hc_ctx();
    c->device = device; c->nq = nq; c->np = np;
    // no configuration from the environment: a cgo host would inherit whatever its shell had. Every switch is an hc_set_option (small32, rot_fuse, pack32, async_alloc);
    // this repo's CLI translates its HCONV_* variables into those calls (host/hconv_host.cpp: applyEnvOptions)
    hipError_t se = hipSetDevice(device);
    if (se == hipSuccess) se = hipStreamCreate(&c->stream);
    if (se != hipSuccess) { delete c; return hc_fail(nullptr, HC_ERR_HIP, "hc_ctx_create: cannot create stream on device %d", device); }
    hipEventCreate(&c->t0); hipEventCreate(&c->t1);
    c->mods.resize((size_t)(nq + np));
    for (int i = 0; i < nq + np; i++) {
        const u64 qi = i < nq ? q[i] : p[i - nq];
        if (qi >> 61 || !h_is_prime(qi) || (qi - 1) % (2ull * HC_N)) {     // the lazy butterflies keep values below 8q: 8q < 2^64 (hc_kernels.h, HC_FM_ALT)
            std::string e = "hc_ctx_create: modulus is not an NTT-friendly prime below 2^61";
            hc_ctx_destroy(c); g_create_err = e; return HC_ERR_ARG;
        }
        HcModHost &mh = c->mods[(size_t)i];
        mh.m.q = qi;
        u64 inv = 1; for (int k = 0; k < 6; k++) inv *= 2 - qi * inv;
        mh.m.qinv = inv;
        u64 r = (u64)((((u128)1) << 64) % qi);
        mh.m.r2 = h_mulmod(r, r, qi);
        mh.m.ninv = h_inv(HC_N, qi); mh.m.ninv_s = h_shoup(mh.m.ninv, qi);
        mh.m.mu = (u64)((((u128)1) << 64) / qi);
        mh.m.row32 = (c->pack32 == 2 && qi < (1ull << 31)) ? 1 : 0;
        u64 g = h_primitive_root(qi), power = (qi - 1) / (2ull * HC_N);
        mh.psi = h_powmod(g, power, qi); mh.psi_inv = h_powmod(g, (qi - 1) - power, qi);
        int rc = hc_build_tables(c, &mh, false); if (!rc) rc = hc_build_tables(c, &mh, true);
        if (rc) { g_create_err = c->err; hc_ctx_destroy(c); return rc; }
    }
    {   // device table of moduli for the leveled (all-limbs-in-one-launch) kernels
        std::vector<HcMod> hm; for (auto &mh : c->mods) hm.push_back(mh.m);
        if (hcx_malloc(c, (void **)&c->d_mods, hm.size() * sizeof(HcMod)) != hipSuccess || hcx_malloc(c, (void **)&c->d_csts, hm.size() * sizeof(HcTw)) != hipSuccess ||
            hcx_h2d(c, c->d_mods, hm.data(), hm.size() * sizeof(HcMod)) != hipSuccess) { g_create_err = "hc_ctx_create: device modulus table"; hc_ctx_destroy(c); return HC_ERR_HIP; }
    }
    if (hcx_malloc(c, (void **)&c->d_rowmods, c->mods.size() * sizeof(HcRowMod)) != hipSuccess || hc_upload_rowmods(c) != HC_OK) { g_create_err = "hc_ctx_create: device table of transforms"; hc_ctx_destroy(c); return HC_ERR_HIP; }
    *out = c;
    return HC_OK;
}

// Releases everything the context owns. void by ABI; a failing HIP call does not stop the teardown, but the first one is kept where
// hc_last_error(NULL) finds it (and reported on stderr): a leak or a sticky device error must not pass silently.
extern "C" void hc_ctx_destroy(hc_ctx *c) {
    if (!c) return;
    hipError_t first = hipSuccess; const char *what = nullptr;
    auto D = [&](hipError_t e, const char *w) { if (e != hipSuccess && first == hipSuccess) { first = e; what = w; } };
    auto F = [&](void *d) { if (d) D(hipFree(d), "hipFree"); };
    D(hipSetDevice(c->device), "hipSetDevice");
    if (c->stream) D(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    for (auto &r : c->prof) { D(hipEventDestroy(r.a), "hipEventDestroy"); D(hipEventDestroy(r.b), "hipEventDestroy"); }
    for (auto &kv : c->cache_free) for (void *d : kv.second) F(d);      // blocks parked by the caching allocator
    for (auto &kv : c->cache_blk) if (kv.second.ev) D(hipEventDestroy(kv.second.ev), "hipEventDestroy");
    for (auto &mh : c->mods) for (void *d : mh.allocs) F(d);
    for (auto &kv : c->evk) { F(kv.second.q_rows); F(kv.second.p_rows); }
    for (auto &kv : c->swk) F(kv.second.rows);
    F(c->idx_pairs); F(c->ws_cts); F(c->ws_cts2); F(c->ws_gather);
    if (c->ev_fork) D(hipEventDestroy(c->ev_fork), "hipEventDestroy");
    if (c->ev_shard) D(hipEventDestroy(c->ev_shard), "hipEventDestroy");
    F(c->enc_roots); F(c->enc_rot_group);
    F(c->ws_ctc); F(c->ws_tmp); F(c->d_mods); F(c->d_rowmods); F(c->ws_mm); F(c->ws_accm);
    for (auto &kv : c->ks_plan) { F(kv.second.bx); F(kv.second.bxdown); F(kv.second.pinv); F(kv.second.pmod); F(kv.second.pinv_qlinv); }
    for (auto &kv : c->rescale_plan) F(kv.second);
    F(c->d_csts);
    if (c->t0) D(hipEventDestroy(c->t0), "hipEventDestroy");
    if (c->t1) D(hipEventDestroy(c->t1), "hipEventDestroy");
    if (c->stream) D(hipStreamDestroy(c->stream), "hipStreamDestroy");
    if (first != hipSuccess) { hc_fail(nullptr, HC_ERR_HIP, "hc_ctx_destroy: %s: %s", what, hipGetErrorString(first)); fprintf(stderr, "libhconv: %s\n", g_create_err.c_str()); }
    delete c;
}

// ------------------------------------------------------------------ memory
extern "C" int hc_malloc(hc_ctx *c, size_t bytes, void **dptr) { HC_ENTER(c); if (!dptr) return hc_fail(c, HC_ERR_ARG, "hc_malloc: null"); HC_HIP(c, hcx_malloc(c, dptr, bytes)); c->allocs_live++; return HC_OK; }
// hipFree drains the device by itself. With cached allocations (option async_alloc = 1) the block is parked for reuse by THIS context, and
// hc_free does not wait for the stream. Plain mode: hipFree drains the device itself. Cached mode (option async_alloc = 1): the block is parked behind an event
// and handed out again only to work queued on this same stream (hcx_free / hcx_h2d_async). What the caller owes: a block is freed into the context that
// allocated it, after every OTHER context that was handed the pointer has been waited for (hc_sync) - the resnet host does both at its two hand-overs
// (hconv_resnet.cpp evalConv_BNRelu_new, hconv_relu.cpp evalConv_BNRelu_tail). Round 2 kept a stream synchronisation here because dropping it broke
// `resnet ... true` under cached allocations: the layer output was a bootstrapper-context block released into the convolution context.
extern "C" int hc_free(hc_ctx *c, void *dptr) {
    HC_ENTER(c);
    hipError_t e = hcx_free(c, dptr);
    if (e == hipErrorInvalidDevicePointer) return hc_fail(c, HC_ERR_ARG, "hc_free: %p was not allocated by this context (with cached allocations a block goes back to the context it came from)", dptr);
    HC_HIP(c, e);
    if (dptr) c->allocs_live--;
    return HC_OK;
}
extern "C" int hc_upload(hc_ctx *c, void *dst, const void *src, size_t bytes) {
    HC_ENTER(c); if (!dst || !src) return hc_fail(c, HC_ERR_ARG, "hc_upload: null pointer");
    HC_HIP(c, hcx_h2d_async(c, dst, src, bytes));
    HC_HIP(c, hipStreamSynchronize(c->stream));   // the host buffer may be released right after return (cgo rule)
    return HC_OK;
}
extern "C" int hc_download(hc_ctx *c, void *dst, const void *src, size_t bytes) {
    HC_ENTER(c); if (!dst || !src) return hc_fail(c, HC_ERR_ARG, "hc_download: null pointer");
    HC_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HC_HIP(c, hipStreamSynchronize(c->stream));
    return HC_OK;
}
extern "C" int hc_copy(hc_ctx *c, void *dst, const void *src, size_t bytes) {
    HC_ENTER(c); if (!dst || !src) return hc_fail(c, HC_ERR_ARG, "hc_copy: null pointer");
    HC_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return HC_OK;
}
extern "C" int hc_device_count(int *n) { if (!n) return HC_ERR_ARG; return hipGetDeviceCount(n) == hipSuccess ? HC_OK : HC_ERR_HIP; }
// device-to-device copy between two contexts' devices (xGMI between GPUs of a node), queued on dst's stream after everything src has queued
extern "C" int hc_copy_peer(hc_ctx *dst_ctx, void *dst, hc_ctx *src_ctx, const void *src, size_t bytes) {
    HC_ENTER(src_ctx);
    if (!dst_ctx || !dst || !src) return hc_fail(src_ctx, HC_ERR_ARG, "hc_copy_peer: null");
    if (!src_ctx->ev_shard) HC_HIP(src_ctx, hipEventCreateWithFlags(&src_ctx->ev_shard, hipEventDisableTiming));
    HC_HIP(src_ctx, hipEventRecord(src_ctx->ev_shard, src_ctx->stream));
    HC_ENTER(dst_ctx);
    HC_HIP(dst_ctx, hipStreamWaitEvent(dst_ctx->stream, src_ctx->ev_shard, 0));
    HC_HIP(dst_ctx, hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device, bytes, dst_ctx->stream));
    return HC_OK;
}
extern "C" int hc_sync(hc_ctx *c) { HC_ENTER(c); HC_HIP(c, hipStreamSynchronize(c->stream)); return hc_prof_flush(c); }

// ------------------------------------------------------------------ L0
static int hc_check_mod(hc_ctx *c, int mod, const char *fn) {
    if (mod < 0 || mod >= c->nq + c->np) return hc_fail(c, HC_ERR_ARG, "%s: modulus index %d out of range", fn, mod);
    return HC_OK;
}
static dim3 hc_pw_grid(size_t n) { size_t b = (n + HC_TPB - 1) / HC_TPB; if (b > 4096) b = 4096; return dim3((unsigned)b); }

static int hc_ensure_tmp(hc_ctx *c, size_t rows) {
    if (c->ws_tmp_rows >= rows) return HC_OK;
    HC_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ws_tmp) HC_HIP(c, hcx_free(c, c->ws_tmp));
    c->ws_tmp = nullptr; c->ws_tmp_rows = 0;
    HC_HIP(c, hcx_malloc(c, (void **)&c->ws_tmp, rows * HC_N * sizeof(u64)));
    c->ws_tmp_rows = rows;
    return HC_OK;
}

extern "C" int hc_ntt(hc_ctx *c, int mod, const uint64_t *in, uint64_t *out, int count) {
    HC_ENTER(c); HC_TRY(hc_check_mod(c, mod, "hc_ntt"));
    if (!in || !out || count < 1) return hc_fail(c, HC_ERR_ARG, "hc_ntt: bad arguments");
    HcModHost &mh = c->mods[(size_t)mod];
    HC_TRY(hc_ensure_tmp(c, (size_t)count));
    HC_TRY(HC_LAUNCH_FM(mh.m.q, c, "cols_fwd", hc_k_cols_fwd, dim3(16, (unsigned)count), (const u64 *)in, c->ws_tmp, mh.fwd, mh.m.q));
    HC_TRY(HC_LAUNCH_FM(mh.m.q, c, "rows_fwd_canon", hc_k_rows_fwd_canon, dim3(16, (unsigned)count), (const u64 *)c->ws_tmp, (u64 *)out, mh.fwd, mh.m.q, mh.m.mu));
    return HC_OK;
}
extern "C" int hc_intt(hc_ctx *c, int mod, const uint64_t *in, uint64_t *out, int count) {
    HC_ENTER(c); HC_TRY(hc_check_mod(c, mod, "hc_intt"));
    if (!in || !out || count < 1) return hc_fail(c, HC_ERR_ARG, "hc_intt: bad arguments");
    HcModHost &mh = c->mods[(size_t)mod];
    HC_TRY(hc_ensure_tmp(c, (size_t)count));
    HC_TRY(hc_launch(c, "rows_inv", hc_k_rows_inv, dim3(16, (unsigned)count), (const u64 *)in, c->ws_tmp, mh.inv, mh.m.q));
    HC_TRY(hc_launch(c, "cols_inv_canon", hc_k_cols_inv_canon, dim3(16, (unsigned)count), (const u64 *)c->ws_tmp, (u64 *)out, mh.inv, mh.m.q));
    return HC_OK;
}
template <int OP>
static int hc_pw(hc_ctx *c, const char *fn, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count, HcTw cst) {
    HC_ENTER(c); HC_TRY(hc_check_mod(c, mod, fn));
    if (!a || !out || count < 1) return hc_fail(c, HC_ERR_ARG, "%s: bad arguments", fn);
    size_t n = (size_t)count * HC_N;
    return hc_launch(c, fn, hc_k_pointwise<OP>, hc_pw_grid(n), (const u64 *)a, (const u64 *)b, (u64 *)out, n, c->mods[(size_t)mod].m, cst);
}
extern "C" int hc_mul(hc_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count) { HcTw z; z.w = z.ws = 0; if (!b) return HC_ERR_ARG; return hc_pw<HC_PW_MUL>(c, "hc_mul", mod, a, b, out, count, z); }
extern "C" int hc_add(hc_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count) { HcTw z; z.w = z.ws = 0; if (!b) return HC_ERR_ARG; return hc_pw<HC_PW_ADD>(c, "hc_add", mod, a, b, out, count, z); }
extern "C" int hc_sub(hc_ctx *c, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count) { HcTw z; z.w = z.ws = 0; if (!b) return HC_ERR_ARG; return hc_pw<HC_PW_SUB>(c, "hc_sub", mod, a, b, out, count, z); }
extern "C" int hc_mul_const(hc_ctx *c, int mod, const uint64_t *a, uint64_t k, uint64_t *out, int count) {
    if (!c) return HC_ERR_ARG;
    HC_TRY(hc_check_mod(c, mod, "hc_mul_const"));
    u64 q = c->mods[(size_t)mod].m.q;
    return hc_pw<HC_PW_MULC>(c, "hc_mul_const", mod, a, a, out, count, h_pair(k % q, q));
}
extern "C" int hc_permute(hc_ctx *c, uint64_t galEl, const uint64_t *in, uint64_t *out, int count) {
    HC_ENTER(c);
    if (!in || !out || in == out || count < 1 || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_permute: bad arguments (in/out must differ, galEl odd)");
    return hc_launch(c, "permute", hc_k_permute, hc_pw_grid((size_t)count * HC_N), (const u64 *)in, (u64 *)out, (u32)(galEl & 0x1FFFF), count);
}

// Under an image batch the images of an operand must not overlap: a stride below the operand's footprint at this level would let the blockIdx.z slices of a launch race
// on shared rows (silent corruption), so it is an argument error. qp: the operand is an extended-basis pair [2][level+1+np][N].
static int hc_batch_fits(hc_ctx *c, const char *fn, int level, bool qp) {
    if (c->nb <= 1) return HC_OK;
    if (c->bs_poly < (size_t)(level + 1) * HC_N) return hc_fail(c, HC_ERR_ARG, "%s: image stride %zu words is below the %d rows of a polynomial at level %d (hc_set_batch)", fn, c->bs_poly, level + 1, level);
    if (qp && c->bs_qp < (size_t)2 * (size_t)(level + 1 + c->np) * HC_N) return hc_fail(c, HC_ERR_ARG, "%s: extended-basis image stride %zu words is below 2 x %d rows at level %d (hc_set_batch)", fn, c->bs_qp, level + 1 + c->np, level);
    return HC_OK;
}
// evaluator.permuteNTT after the key switch (RotateNew / RotateHoisted / ConjugateNew): out0 = Permute(d0 + c0), out1 = Permute(d1), all limbs, one launch
extern "C" int hc_rotate_finish(hc_ctx *c, uint64_t galEl, int level, const uint64_t *d0, const uint64_t *d1, const uint64_t *c0, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq) return hc_fail(c, HC_ERR_ARG, "hc_rotate_finish: level %d outside 0..%d", level, c->nq - 1);
    if (!d0 || !d1 || !c0 || !out0 || !out1 || out0 == d0 || out0 == c0 || out1 == d1 || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_rotate_finish: bad arguments (outputs must differ from inputs, galEl odd)");
    HC_TRY(hc_batch_fits(c, "hc_rotate_finish", level, false));
    return hc_launch(c, "rotate_finish", hc_k_rotate_finish, dim3(HC_GX_LV, (unsigned)(level + 1), 2u * (unsigned)c->nb), (const u64 *)d0, (const u64 *)d1, (const u64 *)c0, (u64 *)out0, (u64 *)out1, (const HcMod *)c->d_mods, (u32)(galEl & 0x1FFFF), c->bs_poly);
}
// ring.PermuteNTTWithIndexLvl on a polynomial at `level` (rows 0..level) / on an extended-basis pair [2][level+1+np][N], for every image of the batch
extern "C" int hc_lv_permute(hc_ctx *c, uint64_t galEl, int level, const uint64_t *in, uint64_t *out) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq || !in || !out || in == out || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_lv_permute: bad arguments (in/out must differ, galEl odd)");
    HC_TRY(hc_batch_fits(c, "hc_lv_permute", level, false));
    return hc_launch(c, "permute", hc_k_permute_mm, dim3(HC_GX_LV, (unsigned)(level + 1), (unsigned)c->nb), (const u64 *)in, (u64 *)out, (u32)(galEl & 0x1FFFF), c->bs_poly, (const HcMod *)c->d_mods, level + 1, c->nq, level + 1);
}
extern "C" int hc_qp_permute2(hc_ctx *c, uint64_t galEl, int level, const uint64_t *in, uint64_t *out) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq || c->np < 1 || !in || !out || in == out || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_qp_permute2: bad arguments (in/out must differ, galEl odd)");
    HC_TRY(hc_batch_fits(c, "hc_qp_permute2", level, true));
    return hc_launch(c, "permute", hc_k_permute_mm, dim3(HC_GX_LV, 2u * (unsigned)(level + 1 + c->np), (unsigned)c->nb), (const u64 *)in, (u64 *)out, (u32)(galEl & 0x1FFFF), c->bs_qp, (const HcMod *)c->d_mods, level + 1, c->nq, level + 1 + c->np);
}
// Image batch of the leveled evaluator (include/hconv.h): n images per launch, the images of an operand stride words apart
extern "C" int hc_set_batch(hc_ctx *c, int n, size_t poly_stride_words, size_t qp_stride_words) {
    if (!c) return HC_ERR_ARG;
    if (n < 1 || n > HC_MAXIMG) return hc_fail(c, HC_ERR_ARG, "hc_set_batch: n=%d outside 1..%d", n, HC_MAXIMG);
    if (n > 1 && (poly_stride_words < (size_t)HC_N || (c->np > 0 && qp_stride_words < (size_t)HC_N))) return hc_fail(c, HC_ERR_ARG, "hc_set_batch: strides must cover at least one row");
    if (n != c->nb || poly_stride_words != c->bs_poly || qp_stride_words != c->bs_qp) c->hoist_cx = nullptr;       // a held decomposition belongs to the batch it was taken under
    c->nb = n; c->bs_poly = n > 1 ? poly_stride_words : 0; c->bs_qp = n > 1 ? qp_stride_words : 0;
    return HC_OK;
}

// ---- leveled polynomials: rows 0..level <-> moduli 0..level (a ring.Poly at that level); one launch covers all limbs
static int hc_lv_check(hc_ctx *c, const char *fn, int level, const void *a, const void *out) {
    if (level < 0 || level >= c->nq) return hc_fail(c, HC_ERR_ARG, "%s: level %d outside 0..%d", fn, level, c->nq - 1);
    if (!a || !out) return hc_fail(c, HC_ERR_ARG, "%s: null", fn);
    return hc_batch_fits(c, fn, level, false);
}
// b_shared: the second operand is a plaintext common to every image of the batch (read once per coefficient: one thread does all images)
template <int OP>
static int hc_lv_pw(hc_ctx *c, const char *fn, int level, const uint64_t *a, const uint64_t *b, uint64_t *out, const uint64_t *consts_host, bool b_shared = false) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, fn, level, a, out));
    if ((OP == HC_PW_MUL || OP == HC_PW_ADD || OP == HC_PW_SUB || OP == HC_PW_MAC) && !b) return hc_fail(c, HC_ERR_ARG, "%s: null", fn);
    HcLvConsts K; memset(&K, 0, sizeof K);
    if (OP == HC_PW_MULC || OP == HC_PW_ADDC) {
        if (!consts_host) return hc_fail(c, HC_ERR_ARG, "%s: null constants", fn);
        if (level >= 32) return hc_fail(c, HC_ERR_UNSUPPORTED, "%s: more than 32 limbs", fn);
        for (int l = 0; l <= level; l++) { const u64 q = c->mods[(size_t)l].m.q; K.c[l] = h_pair(consts_host[l] % q, q); }
    }
    const bool inthread = b_shared && c->nb > 1;
    return hc_launch(c, fn, hc_k_lv_pointwise<OP>, dim3(HC_GX_PW, (unsigned)(level + 1), inthread ? 1u : (unsigned)c->nb), (const u64 *)a, (const u64 *)(b ? b : a), (u64 *)out, (const HcMod *)c->d_mods, K, (size_t)0, (size_t)0, (size_t)0, HC_ROW_IS_MOD, 0,
                     1, inthread ? c->nb : 1, c->bs_poly, b_shared ? (size_t)0 : c->bs_poly, c->bs_poly);
}
// the same operations on both polynomials of a ciphertext in one launch (they may live in separate allocations; ONE plaintext for both polynomials and every image is HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN with b1 null or b0)
template <int OP>
static int hc_lv_pw2(hc_ctx *c, const char *fn, int level, const u64 *a0, const u64 *a1, const u64 *b0, const u64 *b1, u64 *o0, u64 *o1, const uint64_t *consts_host, bool plain = false) {
    HC_TRY(hc_lv_check(c, fn, level, a0, o0));
    if (plain) { if (b1 && b1 != b0) return hc_fail(c, HC_ERR_ARG, "%s: a plaintext operand is ONE polynomial (b1 must be null or b0)", fn); b1 = b0; }
    // hc_version() 1 inferred "one plaintext for every image" from b0 == b1; since version 2 the operation says it (HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN). A product whose two
    // second operands are the same polynomial inside an image batch is almost certainly such a legacy call, and would read pt + z * stride past a one-polynomial allocation
    else if ((OP == HC_PW_MUL || OP == HC_PW_MAC) && c->nb > 1 && b0 && b0 == b1)
        return hc_fail(c, HC_ERR_ARG, "%s: b0 == b1 inside an image batch - a plaintext shared by the images is HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN (hc_version() >= 2)", fn);
    if (!a1 || !o1 || ((OP == HC_PW_MUL || OP == HC_PW_ADD || OP == HC_PW_SUB || OP == HC_PW_MAC) && (!b0 || !b1))) return hc_fail(c, HC_ERR_ARG, "%s: null", fn);
    HcLvConsts K; memset(&K, 0, sizeof K);
    if (OP == HC_PW_MULC) {
        if (!consts_host) return hc_fail(c, HC_ERR_ARG, "%s: null constants", fn);
        if (level >= 32) return hc_fail(c, HC_ERR_UNSUPPORTED, "%s: more than 32 limbs", fn);
        for (int l = 0; l <= level; l++) { const u64 q = c->mods[(size_t)l].m.q; K.c[l] = h_pair(consts_host[l] % q, q); }
    }
    const u64 *bb0 = b0 ? b0 : a0, *bb1 = b1 ? b1 : a1;
    const bool b_shared = plain, inthread = b_shared && c->nb > 1;      // one plaintext for both polynomials and every image: said by the operation (HC_LV_MUL_PLAIN), not inferred from b0 == b1
    return hc_launch(c, fn, hc_k_lv_pointwise<OP>, dim3(HC_GX_PW, (unsigned)(level + 1), inthread ? 2u : 2u * (unsigned)c->nb), a0, bb0, o0, (const HcMod *)c->d_mods, K, (size_t)(a1 - a0), (size_t)(bb1 - bb0), (size_t)(o1 - o0), HC_ROW_IS_MOD, 0,
                     2, inthread ? c->nb : 1, c->bs_poly, b_shared ? (size_t)0 : c->bs_poly, c->bs_poly);
}
extern "C" int hc_lv_op2(hc_ctx *c, int op, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *out0, uint64_t *out1, const uint64_t *consts) {
    HC_ENTER(c);
    switch (op) {
        case HC_LV_MUL: return hc_lv_pw2<HC_PW_MUL>(c, "hc_lv_op2(mul)", level, a0, a1, b0, b1, out0, out1, nullptr);
        case HC_LV_ADD: return hc_lv_pw2<HC_PW_ADD>(c, "hc_lv_op2(add)", level, a0, a1, b0, b1, out0, out1, nullptr);
        case HC_LV_SUB: return hc_lv_pw2<HC_PW_SUB>(c, "hc_lv_op2(sub)", level, a0, a1, b0, b1, out0, out1, nullptr);
        case HC_LV_MUL_CONST: return hc_lv_pw2<HC_PW_MULC>(c, "hc_lv_op2(mul_const)", level, a0, a1, nullptr, nullptr, out0, out1, consts);
        case HC_LV_MUL_ACC: return hc_lv_pw2<HC_PW_MAC>(c, "hc_lv_op2(mul_acc)", level, a0, a1, b0, b1, out0, out1, nullptr);
        case HC_LV_MUL_PLAIN: return hc_lv_pw2<HC_PW_MUL>(c, "hc_lv_op2(mul)", level, a0, a1, b0, b1, out0, out1, nullptr, true);
        case HC_LV_MUL_ACC_PLAIN: return hc_lv_pw2<HC_PW_MAC>(c, "hc_lv_op2(mul_acc)", level, a0, a1, b0, b1, out0, out1, nullptr, true);
    }
    return hc_fail(c, HC_ERR_ARG, "hc_lv_op2: unknown operation %d", op);
}
extern "C" int hc_lv_mul(hc_ctx *c, int level, const uint64_t *a, const uint64_t *b, uint64_t *out) { return hc_lv_pw<HC_PW_MUL>(c, "hc_lv_mul", level, a, b, out, nullptr, false); }
extern "C" int hc_lv_mul_acc(hc_ctx *c, int level, const uint64_t *a, const uint64_t *b, uint64_t *acc) { return hc_lv_pw<HC_PW_MAC>(c, "hc_lv_mul_acc", level, a, b, acc, nullptr, false); }
// b = one plaintext for every image of a batch (include/hconv.h "image batches")
extern "C" int hc_lv_mul_plain(hc_ctx *c, int level, const uint64_t *a, const uint64_t *pt, uint64_t *out) { return hc_lv_pw<HC_PW_MUL>(c, "hc_lv_mul", level, a, pt, out, nullptr, true); }
extern "C" int hc_lv_mul_acc_plain(hc_ctx *c, int level, const uint64_t *a, const uint64_t *pt, uint64_t *acc) { return hc_lv_pw<HC_PW_MAC>(c, "hc_lv_mul_acc", level, a, pt, acc, nullptr, true); }
extern "C" int hc_lv_add(hc_ctx *c, int level, const uint64_t *a, const uint64_t *b, uint64_t *out) { return hc_lv_pw<HC_PW_ADD>(c, "hc_lv_add", level, a, b, out, nullptr); }
extern "C" int hc_lv_sub(hc_ctx *c, int level, const uint64_t *a, const uint64_t *b, uint64_t *out) { return hc_lv_pw<HC_PW_SUB>(c, "hc_lv_sub", level, a, b, out, nullptr); }
extern "C" int hc_lv_mul_const(hc_ctx *c, int level, const uint64_t *a, const uint64_t *consts, uint64_t *out) { return hc_lv_pw<HC_PW_MULC>(c, "hc_lv_mul_const", level, a, nullptr, out, consts); }
extern "C" int hc_lv_add_const(hc_ctx *c, int level, const uint64_t *a, const uint64_t *consts, uint64_t *out) { return hc_lv_pw<HC_PW_ADDC>(c, "hc_lv_add_const", level, a, nullptr, out, consts); }
// batched transforms over rows of different moduli (row y <-> modulus y < nl ? y : nq + y - nl); z operands zs words apart, n images is words apart.
// fuse: optional prologue of the first pass (lift_level: Rescale's lift of t, see HcMm) and epilogue of the second (epi_x: (x - result) * epi_mul (+ epi_add))
struct HcMmFuse { const HcBasisExt *ext_bs = nullptr; int ext_rows = 0; int lift_level = 0; const u64 *epi_x = nullptr; size_t epi_x_zs = 0, epi_x_is = 0; const HcTw *epi_mul = nullptr; const u64 *epi_add = nullptr; size_t epi_add_zs = 0, epi_add_is = 0;
                  const u64 *lift_t = nullptr; size_t lift_t_zs = 0, lift_t_is = 0; const HcTw *lift_pmul = nullptr, *epi_add_mul = nullptr;      // lift_t: ModDown + Rescale in one transform (HcMm)
                  bool out_packed = false;       // out is a library-internal array read only by kernels that expect 4-byte rows for the small moduli (the digits of a key switch)
                  bool raw = false; };           // in / out are plain 8-byte rows whatever the context's pack32 (the rows of a switching key while it is being generated)
static int hc_ntt_mm(hc_ctx *c, const u64 *in, u64 *out, int rows, int nl, int skip_lo, int skip_hi, int z, size_t zs_in, size_t zs_out, int z_alpha = 0, int n = 1, size_t is_in = 0, size_t is_out = 0, const char *tag = "ntt",
                     const HcMmFuse *fuse = nullptr) {
    HC_TRY(hc_ensure_tmp(c, (size_t)rows * z * n));
    HcMm A; memset(&A, 0, sizeof A); A.M = c->d_rowmods; A.nl = nl; A.nq = c->nq; A.skip_lo = skip_lo; A.skip_hi = skip_hi; A.z_alpha = z_alpha; A.nz = z; A.mods = c->d_mods;
    char n1[48], n2[48]; snprintf(n1, sizeof n1, "%s:cols_fwd_mm", tag); snprintf(n2, sizeof n2, "%s:rows_fwd_canon_mm", tag);
    const size_t zt = (size_t)rows * HC_N, it = zt * (size_t)z;
    if (rows > 48) return hc_fail(c, HC_ERR_UNSUPPORTED, "batched transform over more than 48 rows");
    A.gap_lo = rows; A.gap_len = 0;                                          // every row is in the grid
    const dim3 grid(16, (unsigned)rows, (unsigned)(z * n));
    A.zs_in = zs_in; A.is_in = is_in; A.zs_out = zt; A.is_out = it;
    if (fuse && fuse->lift_level > 0) { A.lift_level = fuse->lift_level; if (!fuse->lift_t) { A.zs_in = (size_t)HC_N; A.is_in = (size_t)z * HC_N; } }
    if (fuse && fuse->ext_bs) { A.ext_bs = fuse->ext_bs; A.ext_rows = fuse->ext_rows; }
    if (fuse && fuse->lift_t) { A.lift_t = fuse->lift_t; A.lift_t_zs = fuse->lift_t_zs; A.lift_t_is = fuse->lift_t_is; A.lift_pmul = fuse->lift_pmul; }
    // the extension's operand registers are sized by the most source limbs a digit / ModDown can have: the context's number of special primes (hc_basis_ext_tile)
#define HC_COLS_EXT(E) (c->np <= 2 ? hc_launch(c, c->profile ? n1 : "cols_fwd_mm", hc_k_cols_fwd_mm<E, 2>, grid, in, c->ws_tmp, A) : hc_launch(c, c->profile ? n1 : "cols_fwd_mm", hc_k_cols_fwd_mm<E, HC_MAX_NP>, grid, in, c->ws_tmp, A))
    const bool user32 = c->pack32 == 2 && !(fuse && fuse->raw);              // pack32 = 2: the caller's polynomials (a plain input, the output, the epilogue's operands) carry 4-byte rows too
    A.pk_in = user32; A.pk_out = c->pack32;                                  // the seam between the two passes (ws_tmp) never leaves the library
    if (A.ext_bs && A.lift_t) HC_TRY(HC_COLS_EXT(2));
    else if (A.ext_bs) HC_TRY(HC_COLS_EXT(1));
    else HC_TRY(hc_launch(c, c->profile ? n1 : "cols_fwd_mm", hc_k_cols_fwd_mm<0>, grid, in, c->ws_tmp, A));
#undef HC_COLS_EXT
    A.lift_level = 0; A.ext_bs = nullptr; A.lift_t = nullptr; A.zs_in = zt; A.is_in = it; A.zs_out = zs_out; A.is_out = is_out;
    A.pk_in = c->pack32; A.pk_out = ((c->pack32 && fuse && fuse->out_packed) || user32) ? 1 : 0; A.pk_epi = user32;
    if (fuse && fuse->epi_x) { A.epi_x = fuse->epi_x; A.epi_x_zs = fuse->epi_x_zs; A.epi_x_is = fuse->epi_x_is; A.epi_mul = fuse->epi_mul; A.epi_add = fuse->epi_add; A.epi_add_zs = fuse->epi_add_zs; A.epi_add_is = fuse->epi_add_is; A.epi_add_mul = fuse->epi_add_mul; }
    A.xcd = c->xcd_rows; A.nzn = z * n;
    // a second pass of at most about a round of 16-row workgroups: on quarter tiles (hc_k_rows_fwd_canon_mm_s)
    if ((long)16 * rows * z * n <= c->small_mm_wgs) { A.xcd = 0; HC_TRY(hc_launch<HC_STPB>(c, c->profile ? n2 : "rows_fwd_canon_mm", hc_k_rows_fwd_canon_mm_s, dim3(HC_STILES, (unsigned)rows, (unsigned)(z * n)), (const u64 *)c->ws_tmp, out, A)); return HC_OK; }
    HC_TRY(hc_launch(c, c->profile ? n2 : "rows_fwd_canon_mm", hc_k_rows_fwd_canon_mm, A.xcd ? dim3(16u * (unsigned)rows * (unsigned)(z * n)) : grid, (const u64 *)c->ws_tmp, out, A));
    return HC_OK;
}
// out_user: out is a polynomial of the caller's (hc_lv_intt), not one of the library's coefficient-domain scratch arrays (which keep 8-byte rows under every pack32)
// out_scale / out_gap: the result leaves as the y_i rows of a basis extension (HcMm::out_gap, hc_cols_inv_canon_mm_body)
static int hc_intt_mm(hc_ctx *c, const u64 *in, u64 *out, int rows, int nl, int z, size_t zs_in, size_t zs_out, int skip_lo = 0, int skip_hi = 0, int n = 1, size_t is_in = 0, size_t is_out = 0, const char *tag = "intt", bool out_user = false,
                      const HcTw *out_scale = nullptr, int out_gap = 0) {
    HC_TRY(hc_ensure_tmp(c, (size_t)rows * z * n));
    HcMm A; memset(&A, 0, sizeof A); A.M = c->d_rowmods; A.nl = nl; A.nq = c->nq; A.skip_lo = skip_lo; A.skip_hi = skip_hi; A.z_alpha = 0; A.nz = z;
    char n1[48], n2[48]; snprintf(n1, sizeof n1, "%s:rows_inv_mm", tag); snprintf(n2, sizeof n2, "%s:cols_inv_canon_mm", tag);
    const size_t zt = (size_t)rows * HC_N, it = zt * (size_t)z;
    if (rows > 48) return hc_fail(c, HC_ERR_UNSUPPORTED, "batched transform over more than 48 rows");
    // rows in [skip_lo, skip_hi) have nothing to do: not in the grid
    const int s_lo = skip_lo < 0 ? 0 : (skip_lo > rows ? rows : skip_lo), s_hi = skip_hi < s_lo ? s_lo : (skip_hi > rows ? rows : skip_hi);
    const int cnt = rows - (s_hi - s_lo);
    A.gap_lo = s_lo; A.gap_len = s_hi - s_lo;
    const dim3 grid(16, (unsigned)cnt, (unsigned)(z * n));
    A.zs_in = zs_in; A.is_in = is_in; A.zs_out = zt; A.is_out = it; A.xcd = c->xcd_rows; A.nzn = z * n;
    A.pk_in = c->pack32 == 2; A.pk_out = c->pack32;                          // the caller's NTT-domain rows; the seam (ws_tmp)
    // a launch of a few hundred workgroups costs what ONE workgroup takes: quarter tiles (four residues per thread, four times the workgroups) then
    const bool quarter = (long)16 * cnt * z * n <= c->small_mm_wgs;
    const dim3 sgrid(HC_STILES, (unsigned)cnt, (unsigned)(z * n));
    if (quarter) { A.xcd = 0; HC_TRY(hc_launch<HC_STPB>(c, c->profile ? n1 : "rows_inv_mm", hc_k_rows_inv_mm_s, sgrid, in, c->ws_tmp, A)); }
    else HC_TRY(hc_launch(c, c->profile ? n1 : "rows_inv_mm", hc_k_rows_inv_mm, A.xcd ? dim3(16u * (unsigned)cnt * (unsigned)(z * n)) : grid, in, c->ws_tmp, A));
    A.xcd = 0; A.pk_in = c->pack32; A.pk_out = (c->pack32 == 2 && out_user) ? 1 : 0; A.epi_mul = out_scale; A.out_gap = out_gap;
    A.zs_in = zt; A.is_in = it; A.zs_out = zs_out; A.is_out = is_out;
    if (quarter) HC_TRY(hc_launch<HC_STPB>(c, c->profile ? n2 : "cols_inv_canon_mm", hc_k_cols_inv_canon_mm_s, sgrid, (const u64 *)c->ws_tmp, out, A));
    else HC_TRY(hc_launch(c, c->profile ? n2 : "cols_inv_canon_mm", hc_k_cols_inv_canon_mm, grid, (const u64 *)c->ws_tmp, out, A));
    return HC_OK;
}
static int hc_ensure_mm(hc_ctx *c, size_t rows) {
    if (c->ws_mm_rows >= rows) return HC_OK;
    HC_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ws_mm) HC_HIP(c, hcx_free(c, c->ws_mm));
    c->ws_mm = nullptr; c->ws_mm_rows = 0; c->hoist_cx = nullptr;
    HC_HIP(c, hcx_malloc(c, (void **)&c->ws_mm, rows * HC_N * sizeof(u64)));
    c->ws_mm_rows = rows;
    return HC_OK;
}
extern "C" int hc_lv_ntt(hc_ctx *c, int level, const uint64_t *in, uint64_t *out) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, "hc_lv_ntt", level, in, out));
    return hc_ntt_mm(c, in, out, level + 1, level + 1, 0, 0, 1, 0, 0, 0, c->nb, c->bs_poly, c->bs_poly);
}
extern "C" int hc_lv_intt(hc_ctx *c, int level, const uint64_t *in, uint64_t *out) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, "hc_lv_intt", level, in, out));
    return hc_intt_mm(c, in, out, level + 1, level + 1, 1, 0, 0, 0, 0, c->nb, c->bs_poly, c->bs_poly, "intt", true);
}
extern "C" int hc_lv_mul_tensor(hc_ctx *c, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *d0, uint64_t *d1, uint64_t *d2) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, "hc_lv_mul_tensor", level, a0, d0));
    if (!a1 || !b0 || !b1 || !d1 || !d2) return hc_fail(c, HC_ERR_ARG, "hc_lv_mul_tensor: null");
    return hc_launch(c, "lv_tensor", hc_k_lv_tensor, dim3(HC_GX_TEN, (unsigned)(level + 1), (unsigned)c->nb), (const u64 *)a0, (const u64 *)a1, (const u64 *)b0, (const u64 *)b1, (u64 *)d0, (u64 *)d1, (u64 *)d2, (const HcMod *)c->d_mods, c->bs_poly);
}
// evaluatePolyFromPowerBasis' leaf as one launch (hc_k_lv_lincomb): out_k = sum_t consts[t] a_t,k (+ addc on k = 0). consts: HOST [nterms][level+1], addc: HOST [level+1] or null
extern "C" int hc_lv_lincomb2(hc_ctx *c, int level, int nterms, const uint64_t *const *a0, const uint64_t *const *a1, const uint64_t *consts, const uint64_t *addc, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, "hc_lv_lincomb2", level, a0, out0));
    if (!a1 || !consts || !out1 || nterms < 1 || nterms > HC_MAXLIN || level >= 32) return hc_fail(c, HC_ERR_ARG, "hc_lv_lincomb2: bad arguments (1 <= nterms <= %d, at most 32 limbs)", HC_MAXLIN);
    HcLinPtrs P; HcLinConsts K; memset(&P, 0, sizeof P); memset(&K, 0, sizeof K);
    for (int t = 0; t < nterms; t++) {
        if (!a0[t] || !a1[t]) return hc_fail(c, HC_ERR_ARG, "hc_lv_lincomb2: null term %d", t);
        P.a0[t] = (const u64 *)a0[t]; P.a1[t] = (const u64 *)a1[t];
        for (int l = 0; l <= level; l++) { const u64 q = c->mods[(size_t)l].m.q; K.c[t][l] = (u64)((((u128)(consts[(size_t)t * (level + 1) + l] % q)) << 64) % q); }
    }
    if (addc) for (int l = 0; l <= level; l++) K.addc[l] = addc[l] % c->mods[(size_t)l].m.q;
#define HC_LINCOMB(NT) case NT: return hc_launch(c, "lv_lincomb", hc_k_lv_lincomb<NT>, dim3(HC_GX_LIN, (unsigned)(level + 1), 2u * (unsigned)c->nb), P, K, nterms, (u64 *)out0, (u64 *)out1, (const HcMod *)c->d_mods, c->bs_poly)
    switch (nterms) { HC_LINCOMB(1); HC_LINCOMB(2); HC_LINCOMB(3); HC_LINCOMB(4); HC_LINCOMB(5); HC_LINCOMB(6); HC_LINCOMB(7); HC_LINCOMB(8); }
#undef HC_LINCOMB
    return hc_fail(c, HC_ERR_ARG, "hc_lv_lincomb2: %d terms", nterms);
}
extern "C" int hc_lv_mod_raise(hc_ctx *c, int level, const uint64_t *in_q0, uint64_t *out) {
    HC_ENTER(c); HC_TRY(hc_lv_check(c, "hc_lv_mod_raise", level, in_q0, out));
    if ((const void *)in_q0 == (const void *)out) return hc_fail(c, HC_ERR_ARG, "hc_lv_mod_raise: in and out must differ");
    HC_TRY(hc_ensure_mm(c, (size_t)c->nb)); c->hoist_cx = nullptr;
    u64 *t = c->ws_mm;                                                                  // one coefficient row per image
    HC_TRY(hc_intt_mm(c, in_q0, t, 1, 1, 1, 0, 0, 0, 0, c->nb, c->bs_poly, (size_t)HC_N));
    HC_TRY(hc_launch(c, "mod_raise", hc_k_mod_raise, dim3(HC_GX_LV, (unsigned)(level + 1), (unsigned)c->nb), (const u64 *)t, (u64 *)out, (const HcMod *)c->d_mods, c->bs_poly));
    return hc_ntt_mm(c, out, out, level + 1, level + 1, 0, 0, 1, 0, 0, 0, c->nb, c->bs_poly, c->bs_poly);
}

// getConstAndScale + scaleUpExact of the reference's dependency: a float64 constant with a
// fractional part is scaled by float64(Q[level]); the integer is trunc(fl(fl(c*s) + 0.5)) mod q, q - r if negative.
extern "C" uint64_t hc_const_for(double constant, double q_level_f, uint64_t q, double *scale_mult) {
    double scale = 1.0;
    if (constant != 0) { double frac = constant - (double)(int64_t)constant; if (frac != 0) scale = q_level_f; }
    if (scale_mult) *scale_mult = scale;
    const bool neg = constant < 0;
    double x = (neg ? -scale * constant : scale * constant) + 0.5;
    u64 res;
    if (x < 18446744073709551616.0) res = (u64)x % q;
    else {
        int e; double mant = frexp(x, &e);
        u64 mi = (u64)ldexp(mant, 53); int sh = e - 53;
        u64 r = mi % q; for (int i = 0; i < sh; i++) { r += r; if (r >= q) r -= q; }
        res = r;
    }
    return neg ? q - res : res;
}

// ------------------------------------------------------------------ loop A plumbing
// grid of the fused conv kernels: `jobs` x 16 tiles (x batch); which of the two is blockIdx.x follows HC_JOB_FAST (hc_kernels.h)
static dim3 hc_grid(int jobs, int z = 1) { return HC_JOB_FAST ? dim3((unsigned)jobs, 16, (unsigned)z) : dim3(16, (unsigned)jobs, (unsigned)z); }
static HcPtrs hc_ptrs1(const u64 *p) { HcPtrs P; memset(&P, 0, sizeof P); P.p[0] = p; return P; }
static int hc_fill_loopA(hc_ctx *c, HcLoopA *A, const HcPtrs &kers, u64 *cts, size_t cts_stride, int norm) {
    const HcModHost &m0 = c->mods[0], &m1 = c->mods[1];
    A->ctc = c->ws_ctc; A->ker = kers; A->tmp = c->ws_tmp; A->cts = cts; A->i0 = 0; A->norm = norm; A->slot0 = 0; A->slot_step = norm;
    A->cts_stride = cts_stride;
    A->m0 = m0.m; A->m1 = m1.m;
    A->q1inv = h_pair(h_inv(m1.m.q % m0.m.q, m0.m.q), m0.m.q);
    A->h = (m1.m.q - 1) >> 1; A->negh0 = m0.m.q - (A->h % m0.m.q);
    return HC_OK;
}

// ct_in (2x2 rows) of n ciphertexts times per-limb constants -> ws_ctc as Shoup pairs: c' is the fixed operand of the 2B products
// of loop A, so its companion floor(c' * 2^64 / q) is computed once per conv (4 rows per ciphertext)
static int hc_ensure_ctc(hc_ctx *c, size_t nct) {
    if (c->ws_ctc_cts >= nct) return HC_OK;
    HC_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ws_ctc) HC_HIP(c, hcx_free(c, c->ws_ctc));
    c->ws_ctc = nullptr; c->ws_ctc_cts = 0;
    HC_HIP(c, hcx_malloc(c, (void **)&c->ws_ctc, nct * 4 * HC_N * sizeof(HcTw)));
    c->ws_ctc_cts = nct;
    return HC_OK;
}
static int hc_prepare_ctc(hc_ctx *c, const HcPtrs &ct_in, int n, const u64 cst[2]) {
    HC_TRY(hc_ensure_ctc(c, (size_t)n));
    HcCtc K; K.m0 = c->mods[0].m; K.m1 = c->mods[1].m;
    K.c0 = h_pair(cst[0] % K.m0.q, K.m0.q); K.c1 = h_pair(cst[1] % K.m1.q, K.m1.q);
    return hc_launch(c, "ctc", hc_k_ctc_pairs, dim3(64, 4, (unsigned)n), ct_in, c->ws_ctc, K);
}

// loop A over the channels i = i_first + j*i_stride, j < nch, of each of the n ciphertexts of a batch (blockIdx.z); result j goes to
// cts slot i (compact = false) or j of that ciphertext's array (arrays cts_stride words apart)
static int hc_loopA_run_set(hc_ctx *c, const HcPtrs &kers, int n, int i_first, int i_stride, int nch, u64 *cts, size_t cts_stride, bool compact) {
    long chunk = (c->chunk_nodes < 1 ? 1 : c->chunk_nodes) / n; if (chunk < 1) chunk = 1;     // channels per ciphertext per launch
    if (chunk > nch) chunk = nch;
    HC_TRY(hc_ensure_tmp(c, (size_t)n * (size_t)chunk * 2));
    HcLoopA A; HC_TRY(hc_fill_loopA(c, &A, kers, cts, cts_stride, i_stride));
    const HcModHost &m0 = c->mods[0], &m1 = c->mods[1];
    for (int j0 = 0; j0 < nch; j0 += (int)chunk) {
        const int nj = (int)((nch - j0) < chunk ? (nch - j0) : chunk);
        A.i0 = i_first + j0 * i_stride;
        A.slot0 = compact ? j0 : A.i0; A.slot_step = compact ? 1 : i_stride;
        A.njobs = 2 * nj;
        const dim3 grid = hc_grid(2 * nj, n), flat = hc_grid(2 * nj * n);
        if (hc_f64_ok(m1.m.q)) {
            HC_TRY(hc_launch(c, "a1_mul_rowsinv", hc_k_a1<1>, grid, A, m1.inv_f64));
            HC_TRY(hc_fm_free(m0.m.q) ? hc_launch(c, "a2_colsinv_lift_colsfwd", hc_k_a2<HC_FM_FREE, 1>, flat, A, m1.inv_f64, m0.fwd)
                                      : hc_launch(c, "a2_colsinv_lift_colsfwd", hc_k_a2<HC_FM_ALT, 1>, flat, A, m1.inv_f64, m0.fwd));
        } else {
            HC_TRY(hc_launch(c, "a1_mul_rowsinv", hc_k_a1<0>, grid, A, m1.inv));
            HC_TRY(hc_fm_free(m0.m.q) ? hc_launch(c, "a2_colsinv_lift_colsfwd", hc_k_a2<HC_FM_FREE, 0>, flat, A, m1.inv, m0.fwd)
                                      : hc_launch(c, "a2_colsinv_lift_colsfwd", hc_k_a2<HC_FM_ALT, 0>, flat, A, m1.inv, m0.fwd));
        }
        HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "a3_rowsfwd_rescale", hc_k_a3, grid, A, m0.fwd));
    }
    return HC_OK;
}
static int hc_loopA_run(hc_ctx *c, const u64 *ker, int max_ob, int norm, u64 *cts) {
    return hc_loopA_run_set(c, hc_ptrs1(ker), 1, 0, norm, max_ob / norm, cts, 0, false);   // channels i % norm == 0 (conv.go:526)
}

// qL^-1 mod q_i, i < level, built once per level
static int hc_rescale_plan(hc_ctx *c, int level, const HcTw **out) {
    auto it = c->rescale_plan.find(level);
    if (it == c->rescale_plan.end()) {
        const u64 qL = c->mods[(size_t)level].m.q;
        std::vector<HcTw> h((size_t)level);
        for (int i = 0; i < level; i++) { const u64 q = c->mods[(size_t)i].m.q; h[(size_t)i] = h_pair(h_inv(qL % q, q), q); }
        HcTw *d = nullptr; HC_HIP(c, hcx_malloc(c, (void **)&d, h.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_h2d(c, d, h.data(), h.size() * sizeof(HcTw)));
        it = c->rescale_plan.emplace(level, d).first;
    }
    *out = it->second;
    return HC_OK;
}
// hc_div_round_last (level 1 only on this path): reuse loop A with ker = Montgomery one (c' (*) R = c')
static int hc_div_round_last_n(hc_ctx *c, int level, const u64 *x, size_t xs, u64 *out, size_t os, int np) {
    if (level < 1 || level >= c->nq) return hc_fail(c, HC_ERR_ARG, "hc_div_round_last: level %d outside 1..%d", level, c->nq - 1);
    if (!x || !out) return hc_fail(c, HC_ERR_ARG, "hc_div_round_last: null");
    HC_TRY(hc_batch_fits(c, "hc_div_round_last", level, false));
    if (level != 1) {
        // general level (the leveled evaluator of the convReLU chain): InvNTT of the last limb, centred lift into every lower
        // modulus, NTT there, subtract, multiply by qL^-1 -- six launches whatever the level. in == out is allowed.
        const HcTw *qlinv; HC_TRY(hc_rescale_plan(c, level, &qlinv));
        // np polynomials of each of the nb images per launch (blockIdx.z): x, x + xs and out, out + os (distances in words, modulo 2^64); images bs_poly apart
        const int nb = c->nb, nz = np * nb;
        HC_TRY(hc_ensure_mm(c, (size_t)nz * (level + 1)));
        c->hoist_cx = nullptr;                                   // the scratch is shared with the key switch's decomposition
        u64 *t = c->ws_mm;                                         // t[z][N], z = polynomial + np * image
        // InvNTT of the last limb of every polynomial: the multi-modulus kernels over rows 0..level with rows below `level` skipped
        HC_TRY(hc_intt_mm(c, x, t - (size_t)level * HC_N, level + 1, level + 1, np, xs, (size_t)HC_N, 0, level, nb, c->bs_poly, (size_t)np * HC_N, "rescale"));
        // the lift of t into every lower modulus happens where the forward transform reads its input, (x - NTT(lift)) / q_L where it writes its output: two launches,
        // x read once, the result written once (round 3: lift, two transform passes over a scratch, a finishing pass)
        HcMmFuse F; F.lift_level = level; F.epi_x = x; F.epi_x_zs = xs; F.epi_x_is = c->bs_poly; F.epi_mul = qlinv;
        return hc_ntt_mm(c, t, out, level, level, 0, 0, np, (size_t)HC_N, os, 0, nb, (size_t)np * HC_N, c->bs_poly, "rescale", &F);
    }
    if (c->nb > 1) {           // level 1 in a batch: image by image through the fused level-1 path below (not on the batched chain's route: its rescales end at level 1)
        const int nb = c->nb; const size_t bs = c->bs_poly; int rc = HC_OK;
        c->nb = 1;
        for (int g = 0; g < nb && !rc; g++) rc = hc_div_round_last_n(c, level, x + (size_t)g * bs, xs, out + (size_t)g * bs, os, np);
        c->nb = nb;
        return rc;
    }
    if (np != 1) { HC_TRY(hc_div_round_last_n(c, level, x, 0, out, 0, 1)); return hc_div_round_last_n(c, level, x + xs, 0, out + os, 0, 1); }
    // Build a "ciphertext" whose polynomial 0 is x (constants 1) and a kernel plaintext equal to 1 everywhere.
    HC_TRY(hc_ensure_tmp(c, 16));
    u64 *scratch = nullptr; HC_HIP(c, hcx_malloc(c, (void **)&scratch, (size_t)(2 + 4 + 4) * HC_N * sizeof(u64)));
    u64 *one = scratch, *cts = scratch + 2 * HC_N, *xx = scratch + 6 * HC_N;
    std::vector<u64> h((size_t)2 * HC_N, 1);
    int rc = HC_OK;
    if (hcx_h2d_async(c, one, h.data(), h.size() * sizeof(u64)) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess
        || hipMemcpyAsync(xx, x, 2 * HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess
        || hipMemcpyAsync(xx + 2 * HC_N, x, 2 * HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "hc_div_round_last: staging copies failed");
    const u64 ones[2] = {1, 1};
    if (!rc) rc = hc_prepare_ctc(c, hc_ptrs1(xx), 1, ones);
    if (!rc) rc = hc_loopA_run(c, one, 1, 1, cts);
    if (!rc && hipMemcpyAsync(out, cts, HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "hc_div_round_last: result copy failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = hc_fail(c, HC_ERR_HIP, "hc_div_round_last: stream synchronize failed");
    hcx_free(c, scratch);
    return rc;
}
extern "C" int hc_div_round_last(hc_ctx *c, int level, const uint64_t *x, uint64_t *out) { HC_ENTER(c); return hc_div_round_last_n(c, level, x, 0, out, 0, 1); }
// evaluator.Rescale's one drop on both polynomials of a ciphertext (they may live in separate allocations): same residues as two
// hc_div_round_last calls, half the launches -- the last limb's inverse transform is a 16-workgroup launch whose cost is latency
extern "C" int hc_div_round_last2(hc_ctx *c, int level, const uint64_t *x0, const uint64_t *x1, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    if (!x0 || !x1 || !out0 || !out1) return hc_fail(c, HC_ERR_ARG, "hc_div_round_last2: null");
    return hc_div_round_last_n(c, level, x0, (size_t)(x1 - x0), out0, (size_t)(out1 - out0), 2);
}

// ------------------------------------------------------------------ evk / idx / ker loading
static bool hc_perm_row_local(u64 galEl, int top_bits = 4) {
    // ring.PermuteNTTIndex stays inside the 4096-coefficient tile a b5 workgroup holds in LDS iff it fixes the top 4 bits of the
    // destination index (true for 2^j+1 with j >= 5; with j >= 9 it even fixes the top 8 bits, i.e. the 256-coefficient row)
    for (u32 i = 0; i < HC_N; i++) {
        u32 r = h_bitrev16(i), t = (u32)(((galEl * (2ull * r + 1)) & 0x1FFFF) >> 1), s = h_bitrev16(t);
        if ((s >> (16 - top_bits)) != (i >> (16 - top_bits))) return false;
    }
    return true;
}
extern "C" int hc_evk_load(hc_ctx *c, uint64_t galEl, const uint64_t *b_q, const uint64_t *a_q, const uint64_t *b_p, const uint64_t *a_p) {
    HC_ENTER(c);
    if (!b_q || !a_q || !b_p || !a_p || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_evk_load: bad arguments");
    if (c->np != 1) return hc_fail(c, HC_ERR_UNSUPPORTED, "hc_evk_load: level-0 key switching with one special prime (the pack evaluator of main.go:446-456) is what is implemented; np=%d", c->np);
    const HcModHost &m0 = c->mods[0], &mp = c->mods[(size_t)c->nq];
    // Lattigo's stored form IS the Montgomery form the kernels multiply with: the Q rows are taken as they come (times P^-1, below),
    // the P rows are only re-ordered into the lo-local coalesced order hc_k_b3 reads.
    HcScratch S(c);
    u64 *stage = nullptr; HC_HIP(c, S.alloc(&stage, 2 * HC_N * sizeof(u64)));
    HcEvk e; e.q_rows = nullptr; e.p_rows = nullptr; e.row_local = hc_perm_row_local(galEl); e.row256 = e.row_local && hc_perm_row_local(galEl, 8);
    u64 *stageq = nullptr; HC_HIP(c, S.alloc(&stageq, 2 * HC_N * sizeof(u64)));
    HC_HIP(c, S.alloc(&e.q_rows, 2 * HC_N * sizeof(HcTw)));
    HC_HIP(c, S.alloc(&e.p_rows, 2 * HC_N * sizeof(HcTw)));
    HC_HIP(c, hcx_h2d_async(c, stageq, b_q, HC_N * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, stageq + HC_N, a_q, HC_N * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, stage, b_p, HC_N * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, stage + HC_N, a_p, HC_N * sizeof(u64)));
    // Q rows: stored Montgomery form -> plain residues, times P^-1 mod Q0 once, here (ModDown's final division is then already inside
    // b5's product with the key, hc_k_b4/b5), then (w, floor(w*2^64/Q0)) pairs in natural order
    const HcTw pinv = h_pair(h_inv(mp.m.q % m0.m.q, m0.m.q), m0.m.q);
    HcTw z; z.w = z.ws = 0;
    int rc = hc_launch(c, "evk_from_mont", hc_k_pointwise<HC_PW_FROM_MONT>, hc_pw_grid(2 * HC_N), (const u64 *)stageq, (const u64 *)stageq, stageq, (size_t)2 * HC_N, m0.m, z);
    if (!rc) rc = hc_launch(c, "evk_div_p", hc_k_pointwise<HC_PW_MULC>, hc_pw_grid(2 * HC_N), (const u64 *)stageq, (const u64 *)stageq, stageq, (size_t)2 * HC_N, m0.m, pinv);
    if (!rc) rc = hc_launch(c, "make_pairs", hc_k_make_pairs, hc_pw_grid(2 * HC_N), (const u64 *)stageq, e.q_rows, (size_t)2 * HC_N, m0.m.q, 0);
    // P rows: stored Montgomery form -> plain residues -> (w, floor(w*2^64/P)) pairs in lo-local order
    if (!rc) rc = hc_launch(c, "evk_from_mont", hc_k_pointwise<HC_PW_FROM_MONT>, hc_pw_grid(2 * HC_N), (const u64 *)stage, (const u64 *)stage, stage, (size_t)2 * HC_N, mp.m, z);
    // ... times N^-1 mod P: the inverse transform of the P accumulators (hc_k_b4) then runs without its scaling products
    if (!rc) rc = hc_launch(c, "evk_ninv", hc_k_pointwise<HC_PW_MULC>, hc_pw_grid(2 * HC_N), (const u64 *)stage, (const u64 *)stage, stage, (size_t)2 * HC_N, mp.m, h_pair(mp.m.ninv, mp.m.q));
    if (!rc) rc = hc_launch(c, "make_pairs", hc_k_make_pairs, hc_pw_grid(2 * HC_N), (const u64 *)stage, e.p_rows, (size_t)2 * HC_N, mp.m.q, 1);
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = hc_fail(c, HC_ERR_HIP, "hc_evk_load: stream synchronize failed");
    if (rc) return rc;
    S.keep(e.q_rows); S.keep(e.p_rows);
    auto it = c->evk.find(galEl);
    if (it != c->evk.end()) { hcx_free(c, it->second.q_rows); hcx_free(c, it->second.p_rows); }
    c->evk[galEl] = e;
    return HC_OK;
}

extern "C" int hc_idx_load(hc_ctx *c, const uint64_t *idx_host) {
    HC_ENTER(c);
    const HcModHost &m0 = c->mods[0];
    HcScratch S(c);
    u64 *stage = nullptr; HC_HIP(c, S.alloc(&stage, (size_t)HC_LOGN * HC_N * sizeof(u64)));
    if (idx_host) {
        HC_HIP(c, hcx_h2d_async(c, stage, idx_host, (size_t)HC_LOGN * HC_N * sizeof(u64)));
    } else {   // conv.go:248-253: coeffs[1<<i] = 1 -> EncodeCoeffs(scale 1) -> ToNTT, on the device
        HC_HIP(c, hipMemsetAsync(stage, 0, (size_t)HC_LOGN * HC_N * sizeof(u64), c->stream));
        u64 one = 1;
        for (int i = 0; i < HC_LOGN; i++) HC_HIP(c, hcx_h2d_async(c, stage + (size_t)i * HC_N + ((size_t)1 << i), &one, sizeof one));
        HC_HIP(c, hipStreamSynchronize(c->stream));
        u64 *tmp = nullptr; HC_HIP(c, S.alloc(&tmp, (size_t)HC_LOGN * HC_N * sizeof(u64)));
        HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "cols_fwd", hc_k_cols_fwd, dim3(16, HC_LOGN), (const u64 *)stage, tmp, m0.fwd, m0.m.q));
        HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "rows_fwd_canon", hc_k_rows_fwd_canon, dim3(16, HC_LOGN), (const u64 *)tmp, stage, m0.fwd, m0.m.q, m0.m.mu));
    }
    HcTw *pairs = nullptr; HC_HIP(c, S.alloc(&pairs, (size_t)HC_LOGN * HC_N * sizeof(HcTw)));
    HC_TRY(hc_launch(c, "idx_pairs", hc_k_make_pairs, hc_pw_grid((size_t)HC_LOGN * HC_N), (const u64 *)stage, pairs, (size_t)HC_LOGN * HC_N, m0.m.q, 0));
    HC_HIP(c, hipStreamSynchronize(c->stream));
    S.keep(pairs);
    if (c->idx_pairs) HC_HIP(c, hcx_free(c, c->idx_pairs));
    c->idx_pairs = pairs;
    return HC_OK;
}

static int hc_ker_from_device(hc_ctx *c, u64 *d, int max_ob, bool take, hc_ker **out) {
    // kernel plaintexts are used as they are (plain NTT residues, [i][limb][N]): loop A multiplies them by the Shoup pairs of c'
    u64 *dst = d;
    int rc = HC_OK;
    if (!take) {
        HC_HIP(c, hcx_malloc(c, (void **)&dst, (size_t)max_ob * 2 * HC_N * sizeof(u64)));
        if (hipMemcpyAsync(dst, d, (size_t)max_ob * 2 * HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "hc_ker_load: device copy failed");
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "hc_ker_load: stream synchronize failed");
    if (rc) { hcx_free(c, dst); return rc; }      // `take` means the buffer was ours to free as well
    hc_ker *k = new hc_ker(); k->d = dst; k->max_ob = max_ob; *out = k;
    return HC_OK;
}
extern "C" int hc_ker_load(hc_ctx *c, const uint64_t *host, int max_ob, hc_ker **out) {
    HC_ENTER(c);
    if (!host || !out || max_ob < 1 || c->nq < 2) return hc_fail(c, HC_ERR_ARG, "hc_ker_load: bad arguments");
    HcScratch S(c);
    u64 *d = nullptr; HC_HIP(c, S.alloc(&d, (size_t)max_ob * 2 * HC_N * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, d, host, (size_t)max_ob * 2 * HC_N * sizeof(u64)));
    S.keep(d);                                   // hc_ker_from_device owns it from here (and frees it on failure)
    return hc_ker_from_device(c, d, max_ob, true, out);
}
extern "C" int hc_ker_load_device(hc_ctx *c, const uint64_t *dptr, int max_ob, hc_ker **out) {
    HC_ENTER(c);
    if (!dptr || !out || max_ob < 1 || c->nq < 2) return hc_fail(c, HC_ERR_ARG, "hc_ker_load_device: bad arguments");
    return hc_ker_from_device(c, (u64 *)dptr, max_ob, false, out);
}
// prep_Ker (conv.go:487-518) entirely on the device: scatter/round the k^2*B^2 non-zeros, 2 batched NTTs, Montgomery form
extern "C" int hc_prep_ker(hc_ctx *c, const double *ker_in, int ker_len, const double *bn_a, int in_wid, int ker_wid,
                           int real_ib, int real_ob, int norm, double scale, hc_ker **out) {
    HC_ENTER(c);
    if (!ker_in || !bn_a || !out || c->nq < 2 || in_wid < 1 || ker_wid < 1 || real_ib < 1 || real_ob < 1 || norm < 1)
        return hc_fail(c, HC_ERR_ARG, "hc_prep_ker: bad arguments");
    if (HC_N % (in_wid * in_wid)) return hc_fail(c, HC_ERR_ARG, "hc_prep_ker: in_wid^2 must divide N");
    const int max_bat = HC_N / (in_wid * in_wid), k_sz = ker_wid * ker_wid;
    if (ker_len != k_sz * real_ib * real_ob) return hc_fail(c, HC_ERR_ARG, "input size inconsistent!");   // readTxt's panic text (main.go:986)
    if (norm * real_ib > max_bat || norm * real_ob > max_bat) return hc_fail(c, HC_ERR_ARG, "hc_prep_ker: norm*batch exceeds max_bat=%d", max_bat);
    const int adj = (max_bat - 1) + max_bat * (in_wid + 1) * (ker_wid - 1) / 2;
    if (2 * adj > HC_N) return hc_fail(c, HC_ERR_ARG, "hc_prep_ker: kernel too wide for this input width");
    HcScratch S(c);
    double *dk = nullptr, *da = nullptr; u64 *stage = nullptr, *dst = nullptr;
    HC_HIP(c, S.alloc(&dk, (size_t)ker_len * sizeof(double)));
    HC_HIP(c, S.alloc(&da, (size_t)real_ob * sizeof(double)));
    HC_HIP(c, S.alloc(&stage, (size_t)max_bat * 2 * HC_N * sizeof(u64)));
    HC_HIP(c, S.alloc(&dst, (size_t)max_bat * 2 * HC_N * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, dk, ker_in, (size_t)ker_len * sizeof(double)));
    HC_HIP(c, hcx_h2d_async(c, da, bn_a, (size_t)real_ob * sizeof(double)));
    HC_HIP(c, hipMemsetAsync(stage, 0, (size_t)max_bat * 2 * HC_N * sizeof(u64), c->stream));
    HcPrepKer P; P.ker_in = dk; P.bn_a = da; P.stage = stage; P.in_wid = in_wid; P.ker_wid = ker_wid; P.real_ib = real_ib; P.real_ob = real_ob;
    P.norm = norm; P.max_bat = max_bat; P.scale = scale; P.q0 = c->mods[0].m.q; P.q1 = c->mods[1].m.q;
    int rc = hc_launch(c, "prep_ker_scatter", hc_k_prep_ker, hc_pw_grid((size_t)ker_len), P);
    HC_HIP(c, hipStreamSynchronize(c->stream));      // host buffers may go away after return; hc_ntt below may regrow ws_tmp
    for (int l = 0; l < 2 && !rc; l++) rc = hc_ntt(c, l, stage + (size_t)l * max_bat * HC_N, stage + (size_t)l * max_bat * HC_N, max_bat);
    if (!rc) rc = hc_launch(c, "ker_interleave", hc_k_ker_interleave, hc_pw_grid((size_t)max_bat * 2 * HC_N), (const u64 *)stage, dst, max_bat, c->mods[0].m, c->mods[1].m, 0);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "hc_prep_ker: stream synchronize failed");
    if (rc) return rc;
    S.keep(dst);
    hc_ker *k = new hc_ker(); k->d = dst; k->max_ob = max_bat; *out = k;
    return HC_OK;
}
// plain (non-Montgomery) NTT rows of a kernel handle, [max_ob][2][N] to the HOST: what prep_Ker's pl_ker[i].Value.Coeffs hold
extern "C" int hc_ker_download(hc_ctx *c, const hc_ker *k, uint64_t *host_out) {
    HC_ENTER(c); if (!k || !host_out) return hc_fail(c, HC_ERR_ARG, "hc_ker_download: null");
    const size_t n = (size_t)k->max_ob * 2 * HC_N;
    HC_HIP(c, hipMemcpyAsync(host_out, k->d, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(c, hipStreamSynchronize(c->stream));
    return HC_OK;
}
extern "C" void hc_ker_free(hc_ctx *c, hc_ker *k) { if (!k) return; if (c) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); } hcx_free(c, k->d); delete k; }

// ------------------------------------------------------------------ loop B plumbing
static int hc_fill_loopB(hc_ctx *c, HcLoopB *B, const u64 *src, u64 *dst, const HcEvk &e, int logStep, int step, int norm, u64 galEl, int chunk) {
    const HcModHost &m0 = c->mods[0], &mp = c->mods[(size_t)c->nq];
    B->src = src; B->dst = dst;
    B->tmpC = c->ws_tmp; B->tmpE = c->ws_tmp + (size_t)chunk * HC_N; B->tmpT = c->ws_tmp + (size_t)chunk * 3 * HC_N;
    B->idx = c->idx_pairs + (size_t)logStep * HC_N;
    B->evkQ = e.q_rows; B->evkP = e.p_rows;
    B->n0 = 0; B->step = step; B->norm = norm; B->nodes = 1; B->src_stride = 0; B->dst_stride = 0;
    B->m0 = m0.m; B->mp = mp.m;
    B->pmodq = h_pair(mp.m.q % m0.m.q, m0.m.q);
    B->pinv = h_pair(h_inv(mp.m.q % m0.m.q, m0.m.q), m0.m.q);
    B->mu0 = (u64)((((u128)1) << 64) / m0.m.q);
    {   // switch point of v = uint64(float64(y)/float64(P)) over y in [0,P): binary search with the reference's expression
        const double pf = (double)mp.m.q;
        u64 lo = 0, hi = mp.m.q;              // invariant: v(lo-1) == 0 (or lo == 0), v(hi) >= 1 or hi == P
        while (lo < hi) { u64 mid = lo + (hi - lo) / 2; if ((u64)((double)mid / pf) >= 1) hi = mid; else lo = mid + 1; }
        B->vthresh = lo;
    }
    B->gal = (u32)(galEl & 0x1FFFF);
    return HC_OK;
}
// one tree level of each of the n ciphertexts of a batch: nodes i = 0, norm, 2*norm, ... < step; arrays sstride / dstride words apart
static int hc_pack_level(hc_ctx *c, const u64 *src, u64 *dst, size_t sstride, size_t dstride, int n, int step, int logStep, int norm, u64 galEl, const HcPtrs *bias_last, const HcPtrs *outs_last = nullptr) {
    auto it = c->evk.find(galEl);
    if (it == c->evk.end()) return hc_fail(c, HC_ERR_STATE, "pack: no switching key loaded for galEl=%llu (the reference panics in permuteNTT)", (unsigned long long)galEl);
    if (!it->second.row_local) return hc_fail(c, HC_ERR_UNSUPPORTED, "pack: galEl=%llu does not permute inside 4096-coefficient tiles (needs max_cnum <= 4096)", (unsigned long long)galEl);
    if (!c->idx_pairs) HC_TRY(hc_idx_load(c, nullptr));
    const int nodes = (step + norm - 1) / norm;
    long chunkl = (c->chunk_nodes < 1 ? 1 : c->chunk_nodes) / n; if (chunkl < 1) chunkl = 1;          // nodes per ciphertext per launch
    const int chunk = (int)(chunkl < nodes ? chunkl : nodes);
    HC_TRY(hc_ensure_tmp(c, (size_t)n * chunk * 4));
    HcLoopB B; HC_TRY(hc_fill_loopB(c, &B, src, dst, it->second, logStep, step, norm, galEl, n * chunk));
    B.src_stride = sstride; B.dst_stride = dstride;
    if (it->second.row256) B.tmpT = nullptr;      // hc_k_b5m recomputes t2.c1: b1 need not store it
    HcPtrs nobias; memset(&nobias, 0, sizeof nobias);
    const HcModHost &m0 = c->mods[0], &mp = c->mods[(size_t)c->nq];
    for (int n0 = 0; n0 < nodes; n0 += chunk) {
        const int nn = (nodes - n0) < chunk ? (nodes - n0) : chunk;
        B.n0 = n0; B.nodes = nn;
        const dim3 g1 = hc_grid(n * nn), g2 = hc_grid(2 * n * nn);
        if (c->small_levels > 0 && (long)n * nn <= c->small_levels && it->second.row256 && !HC_JOB_FAST) {
            // a level of a few nodes is one partial wave of workgroups and costs the latency of its five kernels: quarter tiles (1024 residues per workgroup, four per
            // thread: hc_kernels.h, "loop B for SMALL tree levels") spread a row over 64 CUs instead of 16. Same tables, same tmp layouts, same bits.
            const HcPtrs pb = bias_last ? *bias_last : nobias, po = outs_last ? *outs_last : nobias;
            B.tmpT = nullptr;
            const dim3 s1(HC_STILES, (unsigned)(n * nn)), s2(HC_STILES, (unsigned)(2 * n * nn)), s3(HC_STILES, (unsigned)(n * nn), 2);
            HC_TRY(hc_launch<HC_STPB>(c, "b1_node_rowsinv", hc_k_sb1, s1, B, m0.inv));
            HC_TRY(hc_launch<HC_STPB>(c, "b2_colsinv_colsfwdP", hc_k_sb2, s1, B, m0.inv, mp.fwd));
            HC_TRY(hc_launch<HC_STPB>(c, "b3_rowsfwdP_mac_rowsinvP", hc_k_sb3, s3, B, mp.fwd, mp.inv));
            HC_TRY(hc_launch<HC_STPB>(c, "b4_colsinvP_modup_colsfwd", hc_k_sb4, s2, B, mp.inv, m0.fwd));
            HC_TRY(hc_launch<HC_STPB>(c, "b5_rowsfwd_moddown_perm_add", hc_k_sb5, s2, B, m0.fwd, pb, po));
            continue;
        }
        HC_TRY(hc_launch(c, "b1_node_rowsinv", hc_k_b1, g1, B, m0.inv));
        HC_TRY(HC_LAUNCH_FM(mp.m.q, c, "b2_colsinv_colsfwdP", hc_k_b2, g1, B, m0.inv, mp.fwd));
        HC_TRY(HC_LAUNCH_FM(mp.m.q, c, "b3_rowsfwdP_mac_rowsinvP", hc_k_b3, g1, B, mp.fwd, mp.inv));
        HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "b4_colsinvP_modup_colsfwd", hc_k_b4, g2, B, mp.inv, m0.fwd));
        const HcPtrs pb = bias_last ? *bias_last : nobias, po = outs_last ? *outs_last : nobias;
#if HC_B5M_LOADER && !defined(HC_EMU)
        if (it->second.row256) HC_TRY((hc_fm_free(m0.m.q) ? hc_launch<HC_LD_TPB>(c, "b5_rowsfwd_moddown_perm_add", hc_k_b5m_ld<HC_FM_FREE>, g1, B, m0.fwd, pb, po)
                                                          : hc_launch<HC_LD_TPB>(c, "b5_rowsfwd_moddown_perm_add", hc_k_b5m_ld<HC_FM_ALT>, g1, B, m0.fwd, pb, po)));   // + a loader wavefront
#else
        if (it->second.row256) HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "b5_rowsfwd_moddown_perm_add", hc_k_b5m, g1, B, m0.fwd, pb, po));      // one workgroup per (node, tile), both polynomials
#endif
        else HC_TRY(HC_LAUNCH_FM(m0.m.q, c, "b5_rowsfwd_moddown_perm_add", hc_k_b5, g2, B, m0.fwd, pb, po));                        // tile-local permutation (2^j + 1, j = 5..8)
    }
    return HC_OK;
}
// conv.go:266-300 on device-resident level-0 ciphertexts, in place (result in slot 0), for each of the n ciphertext arrays of a batch
// outs != null: the root node writes ciphertext z straight to outs->p[z] ([2][N]) instead of slot 0 of its array
static int hc_pack_run(hc_ctx *c, u64 *cts, size_t cstride, int n, int max_cnum, int real_cnum, const HcPtrs *bias, int stride_log2 = 0, const HcPtrs *outs = nullptr) {
    if (max_cnum < 1 || real_cnum < 1 || max_cnum % real_cnum || (max_cnum & (max_cnum - 1)) || (real_cnum & (real_cnum - 1)))
        return hc_fail(c, HC_ERR_ARG, "pack: max_cnum=%d real_cnum=%d must be powers of two", max_cnum, real_cnum);
    const int norm = max_cnum / real_cnum;
    int step = max_cnum / 2, logStep = 0;
    for (int i = step; i > 1; i /= 2) logStep++;
    if (stride_log2 < 0 || logStep + stride_log2 >= HC_LOGN) return hc_fail(c, HC_ERR_ARG, "pack: stride_log2=%d out of range", stride_log2);
    int j = HC_LOGN - logStep - stride_log2;      // slot m stands for global ciphertext index m << stride_log2
    // Tree levels ping-pong between the caller's array and an internal one: a level reads slots i and i+step of
    // `src` and writes slot i of `dst` (i < step), so no kernel ever reads a row another workgroup is writing.
    const size_t pong_one = (size_t)(max_cnum / 2 > 0 ? max_cnum / 2 : 1) * 2, pong_rows = pong_one * (size_t)n;
    if (c->ws_cts2_rows < pong_rows) {
        HC_HIP(c, hipStreamSynchronize(c->stream));
        if (c->ws_cts2) HC_HIP(c, hcx_free(c, c->ws_cts2));
        c->ws_cts2 = nullptr; c->ws_cts2_rows = 0;
        HC_HIP(c, hcx_malloc(c, (void **)&c->ws_cts2, pong_rows * HC_N * sizeof(u64)));
        c->ws_cts2_rows = pong_rows;
    }
    u64 *src = cts, *dst = c->ws_cts2;
    size_t sstride = cstride, dstride = pong_one * HC_N;
    bool bias_done = false, out_done = false;
    while (step >= norm && step >= 1) {
        const bool last = (step / 2 < norm) || step == 1;
        HC_TRY(hc_pack_level(c, src, dst, sstride, dstride, n, step, logStep + stride_log2, norm, (1ull << j) + 1, last ? bias : nullptr, last ? outs : nullptr));
        if (last) { bias_done = true; out_done = outs != nullptr; }
        u64 *t = src; src = dst; dst = t;
        size_t ts = sstride; sstride = dstride; dstride = ts;
        step /= 2; logStep--; j++;
    }
    if (out_done) return HC_OK;
    if (src != cts) for (int z = 0; z < n; z++)      // result -> slot 0 of the caller's array
        HC_TRY(hc_copy_d2d(c, cts + (size_t)z * cstride, src + (size_t)z * sstride, 2 * HC_N * sizeof(u64)));
    if (bias && !bias_done) {   // max_cnum == real_cnum == 1: no tree level ran
        HcTw z0; z0.w = z0.ws = 0;
        for (int z = 0; z < n; z++) if (bias->p[z])
            HC_TRY(hc_launch(c, "bias_add", hc_k_pointwise<HC_PW_ADD>, hc_pw_grid(HC_N), (const u64 *)(cts + (size_t)z * cstride), bias->p[z], cts + (size_t)z * cstride, (size_t)HC_N, c->mods[0].m, z0));
    }
    if (outs) for (int z = 0; z < n; z++) HC_TRY(hc_copy_d2d(c, (void *)outs->p[z], cts + (size_t)z * cstride, 2 * HC_N * sizeof(u64)));
    return HC_OK;
}
extern "C" int hc_pack_ctxts(hc_ctx *c, uint64_t *cts, int max_cnum, int real_cnum) {
    HC_ENTER(c); if (!cts) return hc_fail(c, HC_ERR_ARG, "hc_pack_ctxts: null");
    return hc_pack_run(c, (u64 *)cts, 0, 1, max_cnum, real_cnum, nullptr);
}
extern "C" int hc_pack_ctxts_strided(hc_ctx *c, uint64_t *cts, int count, int stride_log2, const uint64_t *bias) {
    HC_ENTER(c); if (!cts) return hc_fail(c, HC_ERR_ARG, "hc_pack_ctxts_strided: null");
    const HcPtrs bp = hc_ptrs1((const u64 *)bias);
    return hc_pack_run(c, (u64 *)cts, 0, 1, count, count, bias ? &bp : nullptr, stride_log2);
}

// ------------------------------------------------------------------ key switch / rotate at level 0 (L0 API)
extern "C" int hc_rotate_gal_l0(hc_ctx *c, uint64_t galEl, const uint64_t *c0, const uint64_t *c1, uint64_t *o0, uint64_t *o1);
static int hc_ks_common(hc_ctx *c, uint64_t galEl, const uint64_t *c0, const uint64_t *c1, uint64_t *o0, uint64_t *o1, bool rotate) {
    // Expressed through one pack-tree node: with x = 0 (so t1 = y, t2 = y) and y = (c0, c1) the node computes
    // y + RotateGal(y); RotateGal(y) is recovered by subtracting y. Used only by the L0 parity API; the hot path
    // never takes this detour.
    auto it = c->evk.find(galEl);
    if (it == c->evk.end()) return hc_fail(c, HC_ERR_STATE, "no switching key loaded for galEl=%llu", (unsigned long long)galEl);
    u64 *buf = nullptr; HC_HIP(c, hcx_malloc(c, (void **)&buf, (size_t)6 * HC_N * sizeof(u64)));
    u64 *y = buf, *x = buf + 2 * HC_N, *res = buf + 4 * HC_N;
    int rc = HC_OK;
    hipError_t he = hipMemsetAsync(x, 0, 2 * HC_N * sizeof(u64), c->stream);
    if (he == hipSuccess) he = c0 ? hipMemcpyAsync(y, c0, HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) : hipMemsetAsync(y, 0, HC_N * sizeof(u64), c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(y + HC_N, c1, HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream);
    if (he != hipSuccess) rc = hc_fail(c, HC_ERR_HIP, "level-0 key switch: staging copies failed: %s", hipGetErrorString(he));
    if (!rc && !c->idx_pairs) rc = hc_idx_load(c, nullptr);
    const bool local = it->second.row_local;
    if (!rc) {
        if (rotate && local) {
            // slots: y at 0, x = 0 at 1 => one node with step = 1; the idx row is irrelevant because x = 0
            const int chunk = 1;
            rc = hc_ensure_tmp(c, 4);
            HcLoopB B; if (!rc) rc = hc_fill_loopB(c, &B, y, res, it->second, 0, 1, 1, galEl, chunk);
            const HcModHost &m0 = c->mods[0], &mp = c->mods[(size_t)c->nq];
            if (!rc) rc = hc_launch(c, "b1_node_rowsinv", hc_k_b1, hc_grid(1), B, m0.inv);
            if (!rc) rc = HC_LAUNCH_FM(mp.m.q, c, "b2_colsinv_colsfwdP", hc_k_b2, hc_grid(1), B, m0.inv, mp.fwd);
            if (!rc) rc = HC_LAUNCH_FM(mp.m.q, c, "b3_rowsfwdP_mac_rowsinvP", hc_k_b3, hc_grid(1), B, mp.fwd, mp.inv);
            if (!rc) rc = HC_LAUNCH_FM(m0.m.q, c, "b4_colsinvP_modup_colsfwd", hc_k_b4, hc_grid(2), B, mp.inv, m0.fwd);
            HcPtrs nobias; memset(&nobias, 0, sizeof nobias);
            if (!rc) rc = HC_LAUNCH_FM(m0.m.q, c, "b5_rowsfwd_moddown_perm_add", hc_k_b5, hc_grid(2), B, m0.fwd, nobias, nobias);
            HcTw z; z.w = z.ws = 0;      // node result = y + RotateGal(y): subtract y again
            if (!rc) rc = hc_launch(c, "ks_sub", hc_k_pointwise<HC_PW_SUB>, hc_pw_grid(2 * HC_N), (const u64 *)res, (const u64 *)y, res, (size_t)2 * HC_N, c->mods[0].m, z);
            if (!rc && (hipMemcpyAsync(o0, res, HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess
                        || hipMemcpyAsync(o1, res + HC_N, HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)) rc = hc_fail(c, HC_ERR_HIP, "level-0 key switch: result copies failed");
        } else {
            rc = hc_fail(c, HC_ERR_UNSUPPORTED, "level-0 key switch is exposed for Galois elements that permute inside 4096-coefficient tiles (2^j+1, j>=5), the ones pack_ctxts uses");
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = hc_fail(c, HC_ERR_HIP, "level-0 key switch: stream synchronize failed");
    if (hcx_free(c, buf) != hipSuccess && !rc) rc = hc_fail(c, HC_ERR_HIP, "level-0 key switch: free failed");
    return rc;
}
extern "C" int hc_rotate_gal_l0(hc_ctx *c, uint64_t galEl, const uint64_t *c0, const uint64_t *c1, uint64_t *o0, uint64_t *o1) {
    HC_ENTER(c); if (!c0 || !c1 || !o0 || !o1) return hc_fail(c, HC_ERR_ARG, "hc_rotate_gal_l0: null");
    return hc_ks_common(c, galEl, c0, c1, o0, o1, true);
}
extern "C" int hc_keyswitch_l0(hc_ctx *c, uint64_t galEl, const uint64_t *c1, uint64_t *d0, uint64_t *d1) {
    // SwitchKeysInPlace(c1) = Permute_{g^-1}( RotateGal((0, c1)) ): undo the permutation with the inverse element
    HC_ENTER(c); if (!c1 || !d0 || !d1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_l0: null");
    u64 *buf = nullptr; HC_HIP(c, hcx_malloc(c, (void **)&buf, (size_t)2 * HC_N * sizeof(u64)));
    int rc = hc_ks_common(c, galEl, nullptr, c1, buf, buf + HC_N, true);
    if (!rc) {
        u64 twoN = 2ull * HC_N, ginv = 1, b = galEl % twoN;
        for (u64 e = twoN - 1; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN;
        rc = hc_permute(c, ginv, buf, d0, 1);
        if (!rc) rc = hc_permute(c, ginv, buf + HC_N, d1, 1);
    }
    hipStreamSynchronize(c->stream); hcx_free(c, buf);
    return rc;
}

// ------------------------------------------------------------------ general hybrid key switch (L0, any level)
static HcBasisExt hc_make_bx(const std::vector<u64> &src, u64 t) {
    HcBasisExt B; memset(&B, 0, sizeof B);
    B.n = (int)src.size(); B.t = t; B.mu_t = (u64)((((u128)1) << 64) / t);
    u64 smodt = 1;
    for (int i = 0; i < B.n; i++) {
        const u64 si = src[(size_t)i]; B.s[i] = si; B.mu_s[i] = (u64)((((u128)1) << 64) / si);
        u64 hat_si = 1, hat_t = 1;
        for (int j = 0; j < B.n; j++) if (j != i) { hat_si = h_mulmod(hat_si, src[(size_t)j] % si, si); hat_t = h_mulmod(hat_t, src[(size_t)j] % t, t); }
        B.inv[i] = h_pair(h_inv(hat_si, si), si);
        B.hat[i] = h_pair(hat_t, t);
        smodt = h_mulmod(smodt, si % t, t);
    }
    B.smodt = h_pair(smodt, t);
    return B;
}
extern "C" int hc_swk_load(hc_ctx *c, uint64_t key_id, int level, const uint64_t *rows_host) {
    HC_ENTER(c);
    if (!rows_host || level < 0 || level >= c->nq || c->np < 1 || c->np > HC_MAX_NP) return hc_fail(c, HC_ERR_ARG, "hc_swk_load: bad arguments");
    const int nt = level + 1 + c->np, beta = (level + 1 + c->np - 1) / c->np;
    const size_t n = (size_t)beta * 2 * nt * HC_N;
    HcSwk k; k.level = level; k.beta = beta;
    HC_HIP(c, hcx_malloc(c, (void **)&k.rows, n * sizeof(u64)));
    HC_HIP(c, hcx_h2d_async(c, k.rows, rows_host, n * sizeof(u64)));
    if (c->pack32) HC_TRY(hc_launch(c, "pack32_rows", hc_k_pack32_rows, dim3((unsigned)(beta * 2 * nt)), k.rows, (const HcMod *)c->d_mods, level + 1, c->nq, nt));      // the rows of the ~30-bit limbs as 4-byte words (read by the inner products only)
    HC_HIP(c, hipStreamSynchronize(c->stream));
    auto it = c->swk.find(key_id);
    if (it != c->swk.end()) hcx_free(c, it->second.rows);
    c->swk[key_id] = k;
    return HC_OK;
}
// Harness-side key generation on the device (include/hconv.h): one launch samples every row of the key, one batched transform takes the errors to the
// NTT domain, one launch forms b and the stored (Montgomery) form. 3 + 2 launches per key instead of ~10 one-row launches and three uploads per (digit, limb).
static int hc_swk_generate_impl(hc_ctx *c, uint64_t key_id, int level, uint64_t galEl, const uint64_t *sk_ntt, const uint32_t *seed8, bool splitmix, uint64_t sm_seed, const int64_t *e_host) {
    if (!sk_ntt || level < 0 || level >= c->nq || c->np < 1 || c->np > HC_MAX_NP || (galEl && !(galEl & 1))) return hc_fail(c, HC_ERR_ARG, "hc_swk_generate: bad arguments");
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha;
    if (beta > 63 || nt > 62) return hc_fail(c, HC_ERR_UNSUPPORTED, "hc_swk_generate: more than 62 limbs");
    if (!splitmix && (key_id >> 40)) return hc_fail(c, HC_ERR_ARG, "hc_swk_generate: key id %llu does not fit the 40 bits of the ChaCha nonce that tell keys apart", (unsigned long long)key_id);
    HcKeyGen G; memset(&G, 0, sizeof G);
    if (seed8) memcpy(G.key, seed8, sizeof G.key);
    G.id_lo = (u32)key_id; G.id_hi = (u32)(key_id >> 32) & 0xFFu;
    G.relin = galEl == 0; G.nl = nl; G.nq = c->nq; G.nt = nt; G.alpha = alpha; G.beta = beta;
    if (galEl) { const u64 twoN = 2ull * HC_N; u64 ginv = 1, b = galEl % twoN; for (u64 e = twoN - 1; e; e >>= 1, b = (b * b) % twoN) if (e & 1) ginv = (ginv * b) % twoN; G.ginv = (u32)ginv; }
    const size_t n = (size_t)beta * 2 * nt * HC_N;
    HcScratch S(c);
    HcSwk k; k.level = level; k.beta = beta;
    HC_HIP(c, S.alloc(&k.rows, n * sizeof(u64)));
    HcTw *pm = nullptr; HC_HIP(c, S.alloc(&pm, (size_t)nl * sizeof(HcTw)));
    {   std::vector<HcTw> h((size_t)nl);
        for (int l = 0; l < nl; l++) { const u64 q = c->mods[(size_t)l].m.q; u64 r = 1; for (int j = 0; j < alpha; j++) r = h_mulmod(r, c->mods[(size_t)(c->nq + j)].m.q % q, q); h[(size_t)l] = h_pair(r, q); }
        HC_HIP(c, hcx_h2d(c, pm, h.data(), h.size() * sizeof(HcTw)));
    }
    if (splitmix) {
        long long *de = nullptr; HC_HIP(c, S.alloc(&de, (size_t)beta * HC_N * sizeof(long long)));
        HC_HIP(c, hcx_h2d(c, de, e_host, (size_t)beta * HC_N * sizeof(long long)));
        G.splitmix = 1; G.sm_seed = sm_seed; G.e_in = de;
    }
    const dim3 grid(64, (unsigned)nt, (unsigned)beta);
    HC_TRY(hc_launch(c, "swk_sample", hc_k_swk_sample, grid, k.rows, (const HcMod *)c->d_mods, G));
    c->hoist_cx = nullptr;
    { HcMmFuse Fraw; Fraw.raw = true;                                                                          // key rows are 8-byte words until hc_k_pack32_rows below
      HC_TRY(hc_ntt_mm(c, k.rows, k.rows, nt, nl, 0, 0, beta, (size_t)2 * nt * HC_N, (size_t)2 * nt * HC_N, 0, 1, 0, 0, "ntt", &Fraw)); }      // the e rows (component 0 of every digit), all limbs
    HC_TRY(hc_launch(c, "swk_finish", hc_k_swk_finish, grid, k.rows, (const u64 *)sk_ntt, (const HcMod *)c->d_mods, (const HcTw *)pm, G));
    if (c->pack32) HC_TRY(hc_launch(c, "pack32_rows", hc_k_pack32_rows, dim3((unsigned)(beta * 2 * nt)), k.rows, (const HcMod *)c->d_mods, nl, c->nq, nt));
    HC_HIP(c, hipStreamSynchronize(c->stream));
    S.keep(k.rows);
    auto it = c->swk.find(key_id);
    if (it != c->swk.end()) hcx_free(c, it->second.rows);
    c->swk[key_id] = k;
    return HC_OK;
}
extern "C" int hc_swk_generate(hc_ctx *c, uint64_t key_id, int level, uint64_t galEl, const uint64_t *sk_ntt, const uint32_t *seed8) {
    HC_ENTER(c); if (!seed8) return hc_fail(c, HC_ERR_ARG, "hc_swk_generate: null seed");
    return hc_swk_generate_impl(c, key_id, level, galEl, sk_ntt, seed8, false, 0, nullptr);
}
// TEST harness: the key the test oracle's generator (or_gen_swk) derives from `seed` - uniform rows = counter-based splitmix64, the per-digit errors
// e_host[beta][N] (signed, drawn by the caller with the oracle's Box-Muller so that no device libm is involved) - for replaying the oracle's encrypted network in the product host
extern "C" int hc_swk_generate_splitmix(hc_ctx *c, uint64_t key_id, int level, uint64_t galEl, const uint64_t *sk_ntt, uint64_t seed, const int64_t *e_host) {
    HC_ENTER(c); if (!e_host) return hc_fail(c, HC_ERR_ARG, "hc_swk_generate_splitmix: null errors");
    return hc_swk_generate_impl(c, key_id, level, galEl, sk_ntt, nullptr, true, seed, e_host);
}
// constants of every basis extension of a level, built once: digit d -> target limb T, and {P} -> Q limb l
static int hc_ks_plan(hc_ctx *c, int level, const hc_ctx::KsPlan **out) {
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha;
    auto pit = c->ks_plan.find(level);
    if (pit == c->ks_plan.end()) {
        std::vector<HcBasisExt> hb((size_t)beta * nt), hd((size_t)nl); std::vector<HcTw> hp((size_t)nl), hpm((size_t)nl), hpq((size_t)nl);
        const u64 qL = c->mods[(size_t)level].m.q;
        auto modq = [&](int T) { return c->mods[(size_t)(T < nl ? T : c->nq + (T - nl))].m.q; };
        for (int d = 0; d < beta; d++) {
            const int lo = d * alpha, hi = (d + 1) * alpha < nl ? (d + 1) * alpha : nl;
            std::vector<u64> src; for (int i = lo; i < hi; i++) src.push_back(c->mods[(size_t)i].m.q);
            for (int T = 0; T < nt; T++) hb[(size_t)d * nt + T] = hc_make_bx(src, modq(T));
        }
        std::vector<u64> psrc; for (int j = 0; j < alpha; j++) psrc.push_back(c->mods[(size_t)(c->nq + j)].m.q);
        for (int l = 0; l < nl; l++) {
            const u64 q = c->mods[(size_t)l].m.q; u64 pmod = 1; for (u64 pj : psrc) pmod = h_mulmod(pmod, pj % q, q);
            hd[(size_t)l] = hc_make_bx(psrc, q); hp[(size_t)l] = h_pair(h_inv(pmod, q), q);
            hpm[(size_t)l] = h_pair(pmod, q); hpq[(size_t)l] = l < level ? h_pair(h_inv(h_mulmod(pmod, qL % q, q), q), q) : h_pair(0, q);
        }
        std::vector<HcTw> hy((size_t)nt);
        for (int l = 0; l < nl; l++) hy[(size_t)l] = hb[(size_t)(l / alpha) * nt].inv[l % alpha];
        for (int j = 0; j < alpha; j++) hy[(size_t)(nl + j)] = hd[0].inv[j];
        std::vector<HcTw> hy1(hy);                         // yinv1: 1 on the Q rows (ModDown fused with Rescale transforms row `level` beside the P rows: it leaves unscaled)
        for (int l = 0; l < nl; l++) hy1[(size_t)l] = h_pair(1, c->mods[(size_t)l].m.q);
        hc_ctx::KsPlan P;
        HC_HIP(c, hcx_malloc(c, (void **)&P.yinv, hy.size() * sizeof(HcTw))); HC_HIP(c, hcx_h2d(c, P.yinv, hy.data(), hy.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_malloc(c, (void **)&P.yinv1, hy1.size() * sizeof(HcTw))); HC_HIP(c, hcx_h2d(c, P.yinv1, hy1.data(), hy1.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_malloc(c, (void **)&P.bx, hb.size() * sizeof(HcBasisExt))); HC_HIP(c, hcx_malloc(c, (void **)&P.bxdown, hd.size() * sizeof(HcBasisExt))); HC_HIP(c, hcx_malloc(c, (void **)&P.pinv, hp.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_h2d(c, P.bx, hb.data(), hb.size() * sizeof(HcBasisExt)));
        HC_HIP(c, hcx_h2d(c, P.bxdown, hd.data(), hd.size() * sizeof(HcBasisExt)));
        HC_HIP(c, hcx_h2d(c, P.pinv, hp.data(), hp.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_malloc(c, (void **)&P.pmod, hpm.size() * sizeof(HcTw))); HC_HIP(c, hcx_malloc(c, (void **)&P.pinv_qlinv, hpq.size() * sizeof(HcTw)));
        HC_HIP(c, hcx_h2d(c, P.pmod, hpm.data(), hpm.size() * sizeof(HcTw))); HC_HIP(c, hcx_h2d(c, P.pinv_qlinv, hpq.data(), hpq.size() * sizeof(HcTw)));
        pit = c->ks_plan.emplace(level, P).first;
    }
    *out = &pit->second;
    return HC_OK;
}
// scratch of one key switch at `level` for the nb images of the batch, section-major: digits[img][beta][nt] | acc[img][2][nt] | pc[img][2][alpha + 2] | ext[img][2][nl] | yv[img][max(beta, 2)][alpha + 1]
#ifndef HC_MAC_NB
#define HC_MAC_NB 4                  // images per thread of the key switch's inner product at batches above 2 (hc_k_ks_mac_all)
#endif
struct HcKsScratch { u64 *digits, *acc, *pc, *ext, *yv; size_t digits_is, acc_is, pc_is, ext_is; };
static int hc_ks_scratch(hc_ctx *c, int level, HcKsScratch *S) {
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha; const size_t nb = (size_t)c->nb;
    S->digits_is = (size_t)beta * nt * HC_N; S->acc_is = (size_t)2 * nt * HC_N; S->pc_is = (size_t)2 * (alpha + 2) * HC_N; S->ext_is = (size_t)2 * nl * HC_N;
    const size_t yv_rows = (size_t)(beta > 2 ? beta : 2) * (alpha + 1);                // y_i / v rows of the decomposition's digits, later of ModDown's two polynomials
    HC_TRY(hc_ensure_mm(c, nb * ((size_t)beta * nt + 2 * nt + 2 * (alpha + 2) + 2 * nl + yv_rows)));
    S->digits = c->ws_mm; S->acc = S->digits + nb * S->digits_is; S->pc = S->acc + nb * S->acc_is; S->ext = S->pc + nb * S->pc_is; S->yv = S->ext + nb * S->ext_is;
    return HC_OK;
}
// phase 1 (rlwe.KeySwitcher.DecomposeNTT / ring.Decomposer.DecomposeAndSplit): digits[d][T] = the d-th digit of cx extended to limb
// T (Q limbs 0..level, then the P limbs), NTT domain; a digit's own limbs are not written (phase 2 reads cx there)
static int hc_ks_decompose_into(hc_ctx *c, int level, const u64 *cx, const HcKsScratch &S) {
    const hc_ctx::KsPlan *P; HC_TRY(hc_ks_plan(c, level, &P));
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, beta = (nl + alpha - 1) / alpha, nb = c->nb;
    // cxInvNTT, all limbs, leaving the source side of every digit's extension - y_i = x_i (S/s_i)^-1 mod s_i - where the digit's rows go ([digit + beta * image][alpha + 1][N]: limb l
    // at row l + l / alpha); one pass over them adds the v rows; the target side is taken inside the first pass of the digits' forward transforms (blockIdx.z = digit + beta *
    // image): neither the coefficients of cx nor the extended digits exist in memory in the coefficient domain
    const size_t yz = (size_t)(alpha + 1) * HC_N;
    HC_TRY(hc_intt_mm(c, cx, S.yv, nl, nl, 1, 0, 0, 0, 0, nb, c->bs_poly, (size_t)beta * yz, "decomp", false, P->yinv, alpha));
    HC_TRY(hc_launch(c, "decomp:basis_v", hc_k_basis_v, dim3(HC_GX_YV, (unsigned)(beta * nb)), S.yv, alpha + 1, (const HcBasisExt *)P->bx, nt, alpha, beta));
    HcMmFuse F; F.ext_bs = P->bx; F.ext_rows = nt; F.out_packed = true;            // the digits are read by the inner products only (hc_k_ks_mac_all / _multi)
    return hc_ntt_mm(c, S.yv, S.digits, nt, nl, 0, 0, beta, yz, (size_t)nt * HC_N, alpha, nb, (size_t)beta * yz, S.digits_is, "decomp", &F);
}
// the inner product with both components of the key, all images: acc [img][2][nt][N], images acc_is words apart
static int hc_ks_mac(hc_ctx *c, const HcSwk &key, int level, const u64 *cx, const HcKsScratch &S, u64 *acc, size_t acc_is, const HcMacPrep *prep = nullptr) {
    HcMacPrep PR; memset(&PR, 0, sizeof PR); if (prep) PR = *prep;
    const int alpha = c->np, nl = level + 1, nt = nl + alpha;
    const int nb = c->nb, NB = nb <= 1 ? 1 : nb <= 2 ? 2 : HC_MAC_NB;                 // images per thread; more images = more image groups (blockIdx.z), each reading the key once
#define HC_MAC_ALL(NN) hc_launch(c, "ks_mac_all", hc_k_ks_mac_all<NN>, dim3(HC_GX_MAC, (unsigned)nt, (unsigned)((nb + NN - 1) / NN)), (const u64 *)key.rows, cx, c->bs_poly, (const u64 *)S.digits, S.digits_is, acc, acc_is, (const HcMod *)c->d_mods, nl, c->nq, nt, alpha, key.beta, nb, PR, c->pack32 ? 3 : 0)
    return NB == 1 ? HC_MAC_ALL(1) : NB == 2 ? HC_MAC_ALL(2) : NB == 4 ? HC_MAC_ALL(4) : HC_MAC_ALL(8);
#undef HC_MAC_ALL
}
// ring.(*FastBasisExtender).ModDownSplitNTTPQ for the two polynomials acc[2][nt][N] of every image (basis Q_0..Q_level, P_0..P_(np-1), canonical, NTT):
// InvNTT of the P limbs, {P} -> every Q limb, NTT, (acc - ext) * P^-1
static int hc_ks_moddown(hc_ctx *c, int level, const u64 *acc, size_t acc_is, const HcKsScratch &S, u64 *d0, u64 *d1, uint64_t rot_gal, const u64 *rot_c0, const u64 *add0 = nullptr, const u64 *add1 = nullptr) {
    const hc_ctx::KsPlan *P; HC_TRY(hc_ks_plan(c, level, &P));
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, nb = c->nb;
    // InvNTT of the P rows of both components: rows y -> modulus nq + y (nl = 0)
    const size_t yz = (size_t)(alpha + 1) * HC_N;
    HC_TRY(hc_intt_mm(c, acc + (size_t)nl * HC_N, S.yv, alpha, 0, 2, (size_t)nt * HC_N, yz, 0, 0, nb, acc_is, 2 * yz, "moddown", false, P->yinv + nl, 0));      // as the y_i rows of the extension {P} -> Q (hc_ks_decompose_into)
    HC_TRY(hc_launch(c, "moddown:basis_v", hc_k_basis_v, dim3(HC_GX_YV, 2u * (unsigned)nb), S.yv, alpha + 1, (const HcBasisExt *)P->bxdown, nl, 0, 2));
    HcMmFuse F; F.ext_bs = P->bxdown; F.ext_rows = nl;            // {P} -> every Q limb inside the forward transform's first pass
    if (rot_gal) {   // a rotation: + c0 and the permutation ride in ModDown's last pass (d0, d1 = the rotated ciphertext)
        HC_TRY(hc_ntt_mm(c, S.yv, S.ext, nl, nl, 0, 0, 2, yz, (size_t)nl * HC_N, 0, nb, 2 * yz, S.ext_is, "moddown", &F));
        return hc_launch(c, "ks_moddown_rotate_mm", hc_k_ks_moddown_rotate_mm, dim3(HC_GX_ROT, (unsigned)nl, 2u * (unsigned)nb), (const u64 *)acc, (size_t)nt * HC_N, (const u64 *)S.ext, (size_t)nl * HC_N, rot_c0, d0, d1, (const HcMod *)c->d_mods, (const HcTw *)P->pinv, (u32)(rot_gal & 0x1FFFF),
                         acc_is, S.ext_is, c->bs_poly);
    }
    // (acc - NTT(ext)) / P, plus the addend if there is one (relinearisation: d_k + its key-switched part), in the epilogue of the extension's forward transform: d_k written once
    F.epi_x = acc; F.epi_x_zs = (size_t)nt * HC_N; F.epi_x_is = acc_is; F.epi_mul = P->pinv;
    if (add0) { F.epi_add = add0; F.epi_add_zs = (size_t)(add1 - add0); F.epi_add_is = c->bs_poly; }
    return hc_ntt_mm(c, S.yv, d0, nl, nl, 0, 0, 2, yz, (size_t)(d1 - d0), 0, nb, 2 * yz, c->bs_poly, "moddown", &F);
}
// ModDownSplitNTTPQ and the DivRoundByLastModulusNTT behind it as ONE forward transform per limb (relinearisation followed by Rescale: most multiplications of the
// sine and the ReLU polynomials): d_k = Rescale((acc_k - NTT(ext_k)) / P + add_k), level -> level - 1. Both steps subtract a forward transform from the same limb:
// (x - NTT(ext)) / P + add - NTT(lift), all / q_L, = (x - NTT(ext + P lift)) / (P q_L) + add / q_L, exactly (modular arithmetic, canonical residues: the bits of the
// two-step route). The lift needs the last limb's coefficients after ModDown: InvNTT(acc_L / P + add_L) - ext_L / P, the inverse transform riding with the P rows'.
// acc's row `level` is overwritten.
static int hc_ks_moddown_rescale(hc_ctx *c, int level, u64 *acc, size_t acc_is, const HcKsScratch &S, u64 *d0, u64 *d1, const u64 *add0, const u64 *add1, bool prepped = false) {
    const hc_ctx::KsPlan *P; HC_TRY(hc_ks_plan(c, level, &P));
    const HcTw *qlinv; HC_TRY(hc_rescale_plan(c, level, &qlinv));
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, nb = c->nb;
    const size_t tz = (size_t)(alpha + 2) * HC_N;            // pc[z] = [u -> t | y_0 .. y_(alpha-1) | v]: the extension reads its y_i / v rows where the inverse transform left them
    if (!prepped)       // (the inner product of hc_keyswitch_add_rescale leaves row `level` as acc_L / P + add_L already: HcMacPrep)
    HC_TRY(hc_launch(c, "moddown:mdrs_prep", hc_k_mdrs_prep, dim3(HC_GX_YV, 2u * (unsigned)nb), acc, (size_t)nt * HC_N, acc_is, add0, add0 ? (size_t)(add1 - add0) : (size_t)0, c->bs_poly, level, (const HcTw *)P->pinv, (const HcMod *)c->d_mods));
    // InvNTT of row `level` and of the P rows of both components in one pair of launches: pc[z] = [u | the alpha P rows]
    HC_TRY(hc_intt_mm(c, acc, S.pc - (size_t)level * HC_N, nt, nl, 2, (size_t)nt * HC_N, tz, 0, level, nb, acc_is, S.pc_is, "moddown", false, P->yinv1, 0));       // the P rows leave as y_i (hc_ks_decompose_into), row `level` as it is
    // v of the P rows' extension and, with each coefficient's y_i / v in registers, t = u - ext_L / P on the row before them (what hc_k_mdrs_last did in a launch of its own)
    HC_TRY(hc_launch(c, "moddown:basis_yv", hc_k_basis_yv<true, true>, dim3(HC_GX_YV, 2u * (unsigned)nb), (const u64 *)(S.pc + HC_N), (size_t)HC_N, S.pc + HC_N, alpha + 2, (const HcBasisExt *)P->bxdown, nl, tz, 0, 2, S.pc_is,
                     (const HcBasisExt *)(P->bxdown + level), (const HcTw *)(P->pinv + level)));
    HcMmFuse F; F.ext_bs = P->bxdown; F.ext_rows = nl; F.lift_level = level; F.lift_t = S.pc; F.lift_t_zs = tz; F.lift_t_is = S.pc_is; F.lift_pmul = P->pmod;
    F.epi_x = acc; F.epi_x_zs = (size_t)nt * HC_N; F.epi_x_is = acc_is; F.epi_mul = P->pinv_qlinv;
    if (add0) { F.epi_add = add0; F.epi_add_zs = (size_t)(add1 - add0); F.epi_add_is = c->bs_poly; F.epi_add_mul = qlinv; }
    return hc_ntt_mm(c, S.pc + HC_N, d0, level, level, 0, 0, 2, tz, (size_t)(d1 - d0), 0, nb, S.pc_is, c->bs_poly, "moddown", &F);
}
// phase 2: inner product with the key (both components), then ModDownSplitNTTPQ
static int hc_ks_apply_from(hc_ctx *c, const HcSwk &key, int level, const u64 *cx, const HcKsScratch &S, u64 *d0, u64 *d1, uint64_t rot_gal = 0, const u64 *rot_c0 = nullptr, const u64 *add0 = nullptr, const u64 *add1 = nullptr) {
    HC_TRY(hc_ks_mac(c, key, level, cx, S, S.acc, S.acc_is));
    return hc_ks_moddown(c, level, S.acc, S.acc_is, S, d0, d1, rot_gal, rot_c0, add0, add1);
}
static int hc_ks_find(hc_ctx *c, const char *fn, uint64_t key_id, int level, const HcSwk **key, bool qp_operands = false) {
    HC_TRY(hc_batch_fits(c, fn, level, qp_operands));
    auto it = c->swk.find(key_id);
    if (it == c->swk.end()) return hc_fail(c, HC_ERR_STATE, "%s: no switching key %llu loaded", fn, (unsigned long long)key_id);
    if (level != it->second.level) return hc_fail(c, HC_ERR_ARG, "%s: key %llu is loaded for level %d, not %d", fn, (unsigned long long)key_id, it->second.level, level);
    *key = &it->second;
    return HC_OK;
}
extern "C" int hc_keyswitch(hc_ctx *c, uint64_t key_id, int level, const uint64_t *cx, uint64_t *d0, uint64_t *d1) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch", key_id, level, &key));
    if (!cx || !d0 || !d1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch: null");
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    HC_TRY(hc_ks_decompose_into(c, level, cx, S));
    c->hoist_cx = nullptr;                                   // the scratch no longer holds a hoisted decomposition
    return hc_ks_apply_from(c, *key, level, cx, S, d0, d1);
}
// evaluator.Relinearize's tail in the key switch: out_k = a_k + (key switch of cx)_k, the addition inside ModDown's last pass. out may be a (element-wise in place).
extern "C" int hc_keyswitch_add(hc_ctx *c, uint64_t key_id, int level, const uint64_t *cx, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_add", key_id, level, &key));
    if (!cx || !a0 || !a1 || !out0 || !out1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_add: null");
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    HC_TRY(hc_ks_decompose_into(c, level, cx, S));
    c->hoist_cx = nullptr;
    return hc_ks_apply_from(c, *key, level, cx, S, (u64 *)out0, (u64 *)out1, 0, nullptr, (const u64 *)a0, (const u64 *)a1);
}
// hc_keyswitch_add followed by one hc_div_round_last2, as one call: out_k = Rescale(a_k + (key switch of cx)_k) at level - 1 (level >= 2). ModDown and the rescale share
// one forward transform per limb (hc_ks_moddown_rescale); the residues are those of the two calls. out may be a.
extern "C" int hc_keyswitch_add_rescale(hc_ctx *c, uint64_t key_id, int level, const uint64_t *cx, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_add_rescale", key_id, level, &key));
    if (!cx || !a0 || !a1 || !out0 || !out1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_add_rescale: null");
    if (level < 2) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_add_rescale: level %d: the fused rescale needs level >= 2 (use hc_keyswitch_add and hc_div_round_last2)", level);
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    HC_TRY(hc_ks_decompose_into(c, level, cx, S));
    c->hoist_cx = nullptr;
    const hc_ctx::KsPlan *P; HC_TRY(hc_ks_plan(c, level, &P));
    HcMacPrep PR; PR.pinv = P->pinv; PR.add = (const u64 *)a0; PR.add_zs = (size_t)((const u64 *)a1 - (const u64 *)a0); PR.add_is = c->bs_poly;      // acc_L / P + add_L leaves the inner product (hc_ks_moddown_rescale's first step)
    HC_TRY(hc_ks_mac(c, *key, level, cx, S, S.acc, S.acc_is, &PR));
    return hc_ks_moddown_rescale(c, level, S.acc, S.acc_is, S, (u64 *)out0, (u64 *)out1, (const u64 *)a0, (const u64 *)a1, true);
}
// Hoisted key switching (evaluator.RotateHoisted, conv.go:131; the baby steps of a linear transform): the decomposition of cx is
// computed once and kept in the context; every hc_keyswitch_hoisted with the same (cx, level) then only does the inner product with
// ITS key and the ModDown. Results are bit-identical to hc_keyswitch. The decomposition stays valid until the next hc_keyswitch /
// hc_keyswitch_decompose on this context or until cx is overwritten by the caller.
extern "C" int hc_keyswitch_decompose(hc_ctx *c, int level, const uint64_t *cx) {
    HC_ENTER(c);
    if (!cx || level < 0 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_decompose: bad arguments");
    HC_TRY(hc_batch_fits(c, "hc_keyswitch_decompose", level, false));
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    HC_TRY(hc_ks_decompose_into(c, level, cx, S));
    c->hoist_cx = cx; c->hoist_level = level;
    return HC_OK;
}
extern "C" int hc_keyswitch_hoisted(hc_ctx *c, uint64_t key_id, int level, const uint64_t *cx, uint64_t *d0, uint64_t *d1) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_hoisted", key_id, level, &key));
    if (!cx || !d0 || !d1) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_hoisted: null");
    if (c->hoist_cx != cx || c->hoist_level != level) return hc_fail(c, HC_ERR_STATE, "hc_keyswitch_hoisted: no decomposition of this polynomial at level %d is held (call hc_keyswitch_decompose first)", level);
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    return hc_ks_apply_from(c, *key, level, cx, S, d0, d1);
}

// evaluator.RotateNew / ConjugateNew (permuteNTT) as ONE call: key switch of c1 with the key of galEl, + c0, permutation of both polynomials;
// the same residues as hc_keyswitch + hc_rotate_finish. hoisted != 0: uses the decomposition hc_keyswitch_decompose(level, c1) left in the
// context (evaluator.RotateHoisted). Outputs must not alias the inputs.
extern "C" int hc_keyswitch_rotate(hc_ctx *c, uint64_t key_id, uint64_t galEl, int level, const uint64_t *c0, const uint64_t *c1, uint64_t *out0, uint64_t *out1, int hoisted) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_rotate", key_id, level, &key));
    if (!c0 || !c1 || !out0 || !out1 || out0 == c0 || out0 == c1 || out1 == c1 || out1 == c0 || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_rotate: bad arguments (outputs must differ from inputs, galEl odd)");
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    if (hoisted) {
        if (c->hoist_cx != c1 || c->hoist_level != level) return hc_fail(c, HC_ERR_STATE, "hc_keyswitch_rotate: no decomposition of this polynomial at level %d is held (call hc_keyswitch_decompose first)", level);
    } else {
        HC_TRY(hc_ks_decompose_into(c, level, c1, S));
        c->hoist_cx = nullptr;
    }
    return hc_ks_apply_from(c, *key, level, c1, S, out0, out1, galEl, c0);
}

// ---- the key switch in two halves and arithmetic in the extended basis (Lattigo's MultiplyByDiagMatrixBSGS keeps the baby-step rotations,
// their products with the plaintext diagonals and the giant-step sums in QP and divides by P once per giant step; host: Boot::linear_transform_qp)
// hc_keyswitch_qp = rlwe.(*KeySwitcher).SwitchKeysInPlaceNoModDown (hoisted == 0) / KeyswitchHoistedNoModDown (hoisted != 0: the decomposition
// hc_keyswitch_decompose(level, cx) left in the context): acc[2][level+1+np][N], rows Q_0..Q_level then P_0..P_(np-1), canonical, NTT.
extern "C" int hc_keyswitch_qp(hc_ctx *c, uint64_t key_id, int level, const uint64_t *cx, uint64_t *acc, int hoisted) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_qp", key_id, level, &key, true));
    if (!cx || !acc) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_qp: null");
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    if (hoisted) {
        if (c->hoist_cx != cx || c->hoist_level != level) return hc_fail(c, HC_ERR_STATE, "hc_keyswitch_qp: no decomposition of this polynomial at level %d is held (call hc_keyswitch_decompose first)", level);
    } else {
        HC_TRY(hc_ks_decompose_into(c, level, cx, S));
        c->hoist_cx = nullptr;
    }
    return hc_ks_mac(c, *key, level, cx, S, (u64 *)acc, c->bs_qp);
}
// One rotation of MultiplyByDiagMatrixBSGS kept in the extended basis: hc_keyswitch_qp of cx with the key of galEl, + pc0 (P * c0; may be null) on the Q rows of the first
// component, permutation by galEl of all 2 (level+1+np) rows into out (accumulate != 0: added to out) - the inner product lands in the context's scratch and
// one pass applies the rest. The same residues as hc_keyswitch_qp + hc_lv_add + hc_qp_permute2 (+ hc_qp_op2 ADD).
extern "C" int hc_keyswitch_qp_rotate(hc_ctx *c, uint64_t key_id, uint64_t galEl, int level, const uint64_t *pc0, const uint64_t *cx, uint64_t *out, int hoisted, int accumulate) {
    HC_ENTER(c);
    const HcSwk *key; HC_TRY(hc_ks_find(c, "hc_keyswitch_qp_rotate", key_id, level, &key, true));
    if (!cx || !out || !(galEl & 1)) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_qp_rotate: bad arguments (galEl odd)");
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    if (hoisted) {
        if (c->hoist_cx != cx || c->hoist_level != level) return hc_fail(c, HC_ERR_STATE, "hc_keyswitch_qp_rotate: no decomposition of this polynomial at level %d is held (call hc_keyswitch_decompose first)", level);
    } else {
        HC_TRY(hc_ks_decompose_into(c, level, cx, S));
        c->hoist_cx = nullptr;
    }
    HC_TRY(hc_ks_mac(c, *key, level, cx, S, S.acc, S.acc_is));
    const int nl = level + 1, nt = nl + c->np;
    return hc_launch(c, "qp_rotate_finish", hc_k_qp_rotate_finish, dim3(HC_GX_ROT, (unsigned)nt, 2u * (unsigned)c->nb), (const u64 *)S.acc, S.acc_is, (const u64 *)pc0, c->bs_poly, (u64 *)out, c->bs_qp, (const HcMod *)c->d_mods, nl, c->nq, nt, (u32)(galEl & 0x1FFFF), accumulate ? 1 : 0);
}
// The baby steps of a linear transform as ONE call: nrot hoisted rotations of the decomposition hc_keyswitch_decompose(level, cx) left, each = hc_keyswitch_qp_rotate(key_ids[r],
// galEls[r], level, pc0, cx, outs[r], 1, 0). The inner products of up to 16 / images rotations run in one launch (the digits - n x beta x nt rows - are read once for
// all of them instead of once per rotation), then one finishing pass per rotation. Same residues as the single calls.
extern "C" int hc_keyswitch_qp_rotate_many(hc_ctx *c, int nrot, const uint64_t *key_ids, const uint64_t *galEls, int level, const uint64_t *pc0, const uint64_t *cx, uint64_t *const *outs) {
    HC_ENTER(c);
    if (nrot < 1 || !key_ids || !galEls || !cx || !outs) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_qp_rotate_many: bad arguments");
    if (c->hoist_cx != cx || c->hoist_level != level) return hc_fail(c, HC_ERR_STATE, "hc_keyswitch_qp_rotate_many: no decomposition of this polynomial at level %d is held (call hc_keyswitch_decompose first)", level);
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    const int alpha = c->np, nl = level + 1, nt = nl + alpha, nb = c->nb;
    const int NB = nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8, R = NB == 8 ? 2 : NB == 4 ? 4 : 8;
    const size_t acc_is = (size_t)2 * nt * HC_N, acc_rs = acc_is * (size_t)nb;
    if (c->ws_accm_rows < (size_t)R * nb * 2 * nt) {
        HC_HIP(c, hipStreamSynchronize(c->stream));
        if (c->ws_accm) HC_HIP(c, hcx_free(c, c->ws_accm));
        c->ws_accm = nullptr; c->ws_accm_rows = 0;
        HC_HIP(c, hcx_malloc(c, (void **)&c->ws_accm, (size_t)R * nb * 2 * nt * HC_N * sizeof(u64)));
        c->ws_accm_rows = (size_t)R * nb * 2 * nt;
    }
    std::vector<const HcSwk *> keys((size_t)nrot);                           // every rotation is checked before the first launch: an error leaves nothing half done
    for (int r = 0; r < nrot; r++) {
        HC_TRY(hc_ks_find(c, "hc_keyswitch_qp_rotate_many", key_ids[r], level, &keys[(size_t)r], true));
        if (!(galEls[r] & 1) || !outs[r]) return hc_fail(c, HC_ERR_ARG, "hc_keyswitch_qp_rotate_many: rotation %d: galEl must be odd, out non-null", r);
    }
    for (int r0 = 0; r0 < nrot; r0 += R) {
        const int nr = nrot - r0 < R ? nrot - r0 : R;
        HcKeyPtrs K; memset(&K, 0, sizeof K); int beta = 0;
        for (int r = 0; r < nr; r++) { K.k[r] = keys[(size_t)(r0 + r)]->rows; beta = keys[(size_t)(r0 + r)]->beta; }
        const dim3 grid(HC_GX_MACM, (unsigned)nt);
        HcRotFin F; memset(&F, 0, sizeof F);
        if (c->rot_fuse) {             // out_r = Permute_g(acc_r + P c0): element j is stored where g sends it, i.e. at hc_perm_src(j, g^-1 mod 2N)
            for (int r = 0; r < nr; r++) {
                const u32 g = (u32)(galEls[r0 + r] & 0x1FFFF); u32 x = g;
                for (int it = 0; it < 5; it++) x = (x * (2u - g * x)) & 0x1FFFFu;               // Newton: doubles the correct low bits (3 -> 6 -> 12 -> 24)
                F.out[r] = (u64 *)outs[r0 + r]; F.ginv[r] = x;
            }
            F.pc0 = (const u64 *)pc0; F.pc0_is = c->bs_poly; F.out_is = c->bs_qp;
        }
#define HC_MAC_MULTI(RR, NN) hc_launch(c, "ks_mac_multi", beta >= 3 ? (c->rot_fuse ? hc_k_ks_mac_multi<RR, NN, true, true> : hc_k_ks_mac_multi<RR, NN, false, true>) : (c->rot_fuse ? hc_k_ks_mac_multi<RR, NN, true, false> : hc_k_ks_mac_multi<RR, NN, false, false>), grid, K, nr, (const u64 *)cx, c->bs_poly, (const u64 *)S.digits, S.digits_is, c->ws_accm, acc_rs, acc_is, (const HcMod *)c->d_mods, nl, c->nq, nt, alpha, beta, nb, c->pack32 ? 3 : 0, F)
        if (NB == 8) HC_TRY(HC_MAC_MULTI(2, 8)); else if (NB == 4) HC_TRY(HC_MAC_MULTI(4, 4)); else if (NB == 2) HC_TRY(HC_MAC_MULTI(8, 2)); else HC_TRY(HC_MAC_MULTI(8, 1));
#undef HC_MAC_MULTI
        if (!c->rot_fuse)
        for (int r = 0; r < nr; r++)
            HC_TRY(hc_launch(c, "qp_rotate_finish", hc_k_qp_rotate_finish, dim3(HC_GX_ROT, (unsigned)nt, 2u * (unsigned)nb), (const u64 *)(c->ws_accm + (size_t)r * acc_rs), acc_is, (const u64 *)pc0, c->bs_poly, (u64 *)outs[r0 + r], c->bs_qp,
                             (const HcMod *)c->d_mods, nl, c->nq, nt, (u32)(galEls[r0 + r] & 0x1FFFF), 0));
    }
    return HC_OK;
}
// hc_mod_down2 = ring.(*FastBasisExtender).ModDownSplitNTTPQ on the two polynomials x[2][level+1+np][N] -> out0, out1 [level+1][N]. A hoisted
// decomposition held by the context survives it when it was taken at this same level (any other level drops it).
extern "C" int hc_mod_down2(hc_ctx *c, int level, const uint64_t *x, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "hc_mod_down2: level %d outside 0..%d or no special primes", level, c->nq - 1);
    if (!x || !out0 || !out1) return hc_fail(c, HC_ERR_ARG, "hc_mod_down2: null");
    HC_TRY(hc_batch_fits(c, "hc_mod_down2", level, true));
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    if (level != c->hoist_level) c->hoist_cx = nullptr;      // the scratch layout depends on the level: pc / ext of another level overlap the held digits
    return hc_ks_moddown(c, level, (const u64 *)x, c->bs_qp, S, (u64 *)out0, (u64 *)out1, 0, nullptr);
}
// hc_mod_down2 followed by + a_k and ONE hc_div_round_last2, as one call (the end of a linear transform: ModDown of the accumulators, the other terms, Rescale's first drop):
// out_k = Rescale(ModDown(x)_k + a_k) at level - 1, level >= 2; a0 / a1 may be NULL (no addend; both or neither). One forward transform per limb (hc_ks_moddown_rescale); the
// residues of the three calls. Row `level` of both components of x is overwritten.
extern "C" int hc_mod_down2_add_rescale(hc_ctx *c, int level, uint64_t *x, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    if (level < 2 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "hc_mod_down2_add_rescale: level %d outside 2..%d or no special primes", level, c->nq - 1);
    if (!x || !out0 || !out1 || (a0 == nullptr) != (a1 == nullptr)) return hc_fail(c, HC_ERR_ARG, "hc_mod_down2_add_rescale: null (the addends come as a pair)");
    HC_TRY(hc_batch_fits(c, "hc_mod_down2_add_rescale", level, true));
    HcKsScratch S; HC_TRY(hc_ks_scratch(c, level, &S));
    if (level != c->hoist_level) c->hoist_cx = nullptr;
    return hc_ks_moddown_rescale(c, level, (u64 *)x, c->bs_qp, S, (u64 *)out0, (u64 *)out1, (const u64 *)a0, (const u64 *)a1);
}
// hc_qp_op2: out_k = a_k (op) b_k, k = 0, 1, over the level+1+np rows of the extended basis (op: HC_LV_MUL, HC_LV_ADD, HC_LV_MUL_ACC; b1 == b0
// for a plaintext operand; products of two NTT residues as hc_lv_mul)
extern "C" int hc_qp_op2(hc_ctx *c, int op, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *out0, uint64_t *out1) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "hc_qp_op2: level %d outside 0..%d or no special primes", level, c->nq - 1);
    const bool plain = op == HC_LV_MUL_PLAIN || op == HC_LV_MUL_ACC_PLAIN;
    if (plain) { if (b1 && b1 != b0) return hc_fail(c, HC_ERR_ARG, "hc_qp_op2: a plaintext operand is ONE polynomial over the extended basis (b1 must be null or b0)"); b1 = b0; op = op == HC_LV_MUL_PLAIN ? HC_LV_MUL : HC_LV_MUL_ACC; }
    else if ((op == HC_LV_MUL || op == HC_LV_MUL_ACC) && c->nb > 1 && b0 && b0 == b1)      // the legacy form of a shared plaintext (hc_lv_pw2)
        return hc_fail(c, HC_ERR_ARG, "hc_qp_op2: b0 == b1 inside an image batch - a plaintext shared by the images is HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN (hc_version() >= 2)");
    if (!a0 || !a1 || !b0 || !b1 || !out0 || !out1) return hc_fail(c, HC_ERR_ARG, "hc_qp_op2: null");
    HC_TRY(hc_batch_fits(c, "hc_qp_op2", level, true));
    HcLvConsts K; memset(&K, 0, sizeof K);
    const size_t as = (size_t)(a1 - a0), bs = (size_t)(b1 - b0), os = (size_t)(out1 - out0);
    const bool b_shared = plain, inthread = b_shared && c->nb > 1;                                  // a plaintext (an encoded diagonal): one for every image, said by the operation
    const dim3 grid(HC_GX_PW, (unsigned)(level + 1 + c->np), inthread ? 2u : 2u * (unsigned)c->nb);
    const int nin = inthread ? c->nb : 1; const size_t ia = c->bs_qp, ib = b_shared ? (size_t)0 : c->bs_qp, io = c->bs_qp;
    switch (op) {
        case HC_LV_MUL: return hc_launch(c, "hc_qp_op2(mul)", hc_k_lv_pointwise<HC_PW_MUL>, grid, (const u64 *)a0, (const u64 *)b0, (u64 *)out0, (const HcMod *)c->d_mods, K, as, bs, os, level + 1, c->nq, 2, nin, ia, ib, io);
        case HC_LV_ADD: return hc_launch(c, "hc_qp_op2(add)", hc_k_lv_pointwise<HC_PW_ADD>, grid, (const u64 *)a0, (const u64 *)b0, (u64 *)out0, (const HcMod *)c->d_mods, K, as, bs, os, level + 1, c->nq, 2, nin, ia, ib, io);
        case HC_LV_MUL_ACC: return hc_launch(c, "hc_qp_op2(mul_acc)", hc_k_lv_pointwise<HC_PW_MAC>, grid, (const u64 *)a0, (const u64 *)b0, (u64 *)out0, (const HcMod *)c->d_mods, K, as, bs, os, level + 1, c->nq, 2, nin, ia, ib, io);
    }
    return hc_fail(c, HC_ERR_ARG, "hc_qp_op2: unknown operation %d", op);
}

// hc_qp_mul_sum: out (+)= sum_t a_t (*) pt_t over the extended basis - the diagonal sum of a giant step in one launch (hc_k_qp_mul_sum)
extern "C" int hc_qp_mul_sum(hc_ctx *c, int level, int nterms, const uint64_t *const *a, const uint64_t *const *pt, uint64_t *out, int accumulate) {
    HC_ENTER(c);
    if (level < 0 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "hc_qp_mul_sum: level %d outside 0..%d or no special primes", level, c->nq - 1);
    if (!a || !pt || !out || nterms < 1 || nterms > HC_MAXTERMS) return hc_fail(c, HC_ERR_ARG, "hc_qp_mul_sum: bad arguments (1 <= nterms <= %d)", HC_MAXTERMS);
    HC_TRY(hc_batch_fits(c, "hc_qp_mul_sum", level, true));
    HcTermPtrs P; memset(&P, 0, sizeof P);
    for (int t = 0; t < nterms; t++) { if (!a[t] || !pt[t] || a[t] == out) return hc_fail(c, HC_ERR_ARG, "hc_qp_mul_sum: null or aliased term %d", t); P.a[t] = (const u64 *)a[t]; P.pt[t] = (const u64 *)pt[t]; }
    const int nt = level + 1 + c->np;
    return hc_launch(c, "qp_mul_sum", hc_k_qp_mul_sum, dim3(HC_GX_QPMS, (unsigned)nt, 2), P, nterms, (u64 *)out, (const HcMod *)c->d_mods, level + 1, c->nq, nt, c->nb, c->bs_qp, c->bs_qp, accumulate ? 1 : 0);
}

// several giant steps' sums from one pass over the rotations: out[h] (+)= sum_t a[t] (*) pt[h * nterms + t], h < ngiant <= 4; a diagonal pointer may be NULL (giant step h has
// no diagonal for baby step t; every t has at least one). The residues of ngiant hc_qp_mul_sum calls.
static int hc_qp_mul_sum_g(hc_ctx *c, const char *fn, int level, int nterms, int ngiant, const uint64_t *const *a, const uint64_t *const *pt, uint64_t *const *out, const int *accumulate) {
    if (level < 0 || level >= c->nq || c->np < 1) return hc_fail(c, HC_ERR_ARG, "%s: level %d outside 0..%d or no special primes", fn, level, c->nq - 1);
    if (!a || !pt || !out || !accumulate || ngiant < 1 || ngiant > HC_MAXGIANT || nterms < 1 || nterms > HC_MAXTERMS) return hc_fail(c, HC_ERR_ARG, "%s: bad arguments (1 <= nterms <= %d, 1 <= giant steps <= %d)", fn, HC_MAXTERMS, HC_MAXGIANT);
    HC_TRY(hc_batch_fits(c, fn, level, true));
    HcTermPtrsG P; memset(&P, 0, sizeof P);
    for (int h = 0; h < ngiant; h++) {
        if (!out[h]) return hc_fail(c, HC_ERR_ARG, "%s: null output %d", fn, h);
        for (int h2 = 0; h2 < h; h2++) if (out[h2] == out[h]) return hc_fail(c, HC_ERR_ARG, "%s: outputs %d and %d are the same", fn, h2, h);
        P.out[h] = (u64 *)out[h]; P.acc[h] = accumulate[h] ? 1 : 0;
    }
    for (int t = 0; t < nterms; t++) {
        bool any = false;
        for (int h = 0; h < ngiant; h++) { P.pt[h][t] = (const u64 *)pt[(size_t)h * nterms + t]; any = any || P.pt[h][t]; if (a[t] == out[h]) return hc_fail(c, HC_ERR_ARG, "%s: term %d aliases output %d", fn, t, h); }
        if (!a[t] || !any) return hc_fail(c, HC_ERR_ARG, "%s: term %d is null or has no diagonal", fn, t);
        P.a[t] = (const u64 *)a[t];
    }
    const int nt = level + 1 + c->np, nb = c->nb, NB = nb <= 1 ? 1 : nb <= 2 ? 2 : 4;
#define HC_QPMSG(GG, NN) hc_launch(c, "qp_mul_sum_g", hc_k_qp_mul_sum_g<GG, NN>, dim3(HC_GX_QPMS, (unsigned)nt, 2u * (unsigned)((nb + NN - 1) / NN)), P, nterms, (const HcMod *)c->d_mods, level + 1, c->nq, nt, nb, c->bs_qp, c->bs_qp)
#define HC_QPMSG_N(GG) (NB == 1 ? HC_QPMSG(GG, 1) : NB == 2 ? HC_QPMSG(GG, 2) : HC_QPMSG(GG, 4))
    return ngiant == 1 ? HC_QPMSG_N(1) : ngiant == 2 ? HC_QPMSG_N(2) : ngiant == 3 ? HC_QPMSG_N(3) : HC_QPMSG_N(4);
#undef HC_QPMSG_N
#undef HC_QPMSG
}
extern "C" int hc_qp_mul_sum_many(hc_ctx *c, int level, int nterms, int ngiant, const uint64_t *const *a, const uint64_t *const *pt, uint64_t *const *out, const int *accumulate) {
    HC_ENTER(c);
    return hc_qp_mul_sum_g(c, "hc_qp_mul_sum_many", level, nterms, ngiant, a, pt, out, accumulate);
}
extern "C" int hc_qp_mul_sum2(hc_ctx *c, int level, int nterms, const uint64_t *const *a, const uint64_t *const *pt0, const uint64_t *const *pt1, uint64_t *out0, uint64_t *out1, int accumulate0, int accumulate1) {
    HC_ENTER(c);
    if (!pt0 || !pt1 || nterms < 1 || nterms > HC_MAXTERMS) return hc_fail(c, HC_ERR_ARG, "hc_qp_mul_sum2: bad arguments");
    std::vector<const uint64_t *> pt((size_t)2 * nterms); for (int t = 0; t < nterms; t++) { pt[(size_t)t] = pt0[t]; pt[(size_t)nterms + t] = pt1[t]; }
    uint64_t *outs[2] = {out0, out1}; const int acc[2] = {accumulate0, accumulate1};
    return hc_qp_mul_sum_g(c, "hc_qp_mul_sum2", level, nterms, 2, a, pt.data(), outs, acc);
}

// ------------------------------------------------------------------ L1
static int hc_ensure_cts(hc_ctx *c, size_t rows) {
    if (c->ws_cts_rows >= rows) return HC_OK;
    HC_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ws_cts) HC_HIP(c, hcx_free(c, c->ws_cts));
    c->ws_cts = nullptr; c->ws_cts_rows = 0;
    HC_HIP(c, hcx_malloc(c, (void **)&c->ws_cts, rows * HC_N * sizeof(u64)));
    c->ws_cts_rows = rows;
    return HC_OK;
}
// scale bookkeeping of conv.go:527-528 (MulNew, SetScale = MultByConst + Rescale) -> per-limb constants
static int hc_loopA_consts(hc_ctx *c, double ct_scale, double ker_scale, int max_ob, int norm, double out_scale, u64 cst[2], double *target) {
    if (c->nq < 2) return hc_fail(c, HC_ERR_STATE, "conv: needs the level-1 modulus chain (nq >= 2)");
    if (max_ob < 1 || norm < 1 || max_ob % norm) return hc_fail(c, HC_ERR_ARG, "conv: max_ob=%d norm=%d", max_ob, norm);
    const double tgt = out_scale / (double)(max_ob / norm);
    const double prod = ct_scale * ker_scale, constant = tgt / prod;
    double smul = 1;
    for (int l = 0; l < 2; l++) cst[l] = hc_const_for(constant, (double)c->mods[1].m.q, c->mods[(size_t)l].m.q, &smul);
    // Rescale's drop loop (upstream Rescale): exactly one limb must go (level 1 -> 0)
    double sc = prod * smul; int drops = 0, level = 1;
    while (level - drops > 0 && sc / (double)c->mods[(size_t)(level - drops)].m.q >= tgt / 2) { sc /= (double)c->mods[(size_t)(level - drops)].m.q; drops++; }
    if (drops != 1) return hc_fail(c, HC_ERR_STATE, "conv: SetScale would drop %d limbs instead of 1 (scales ct=%g ker=%g out=%g)", drops, ct_scale, ker_scale, out_scale);
    *target = tgt;
    return HC_OK;
}
extern "C" int hc_conv_mult_phase(hc_ctx *c, const uint64_t *ct_in, double ct_scale, const hc_ker *ker, double ker_scale,
                                  int max_ob, int norm, double out_scale, uint64_t *cts_out) {
    HC_ENTER(c);
    if (!ct_in || !ker || !cts_out || ker->max_ob < max_ob) return hc_fail(c, HC_ERR_ARG, "hc_conv_mult_phase: bad arguments");
    u64 cst[2]; double target;
    HC_TRY(hc_loopA_consts(c, ct_scale, ker_scale, max_ob, norm, out_scale, cst, &target));
    HC_TRY(hc_prepare_ctc(c, hc_ptrs1((const u64 *)ct_in), 1, cst));
    return hc_loopA_run(c, ker->d, max_ob, norm, (u64 *)cts_out);
}
// n convolutions (same shape and scales, independent ciphertexts) as ONE launch set: every kernel of loop A and of the pack tree
// covers all n ciphertexts (a grid dimension), the switching keys, idx plaintexts and twiddle tables are shared, and the top tree
// levels (1..16 nodes per ciphertext, latency-bound when launched alone) carry n times the work per launch.
static int hc_conv_batch_run(hc_ctx *c, int n, const HcPtrs &ct_in, const HcPtrs &kers, const HcPtrs &bias, bool any_bias, u64 *const *ct_out,
                             int max_ob, int norm, const u64 cst[2]) {
    HC_TRY(hc_prepare_ctc(c, ct_in, n, cst));
    HC_TRY(hc_ensure_cts(c, (size_t)n * max_ob * 2));
    const size_t cstride = (size_t)max_ob * 2 * HC_N;
    HC_TRY(hc_loopA_run_set(c, kers, n, 0, norm, max_ob / norm, c->ws_cts, cstride, false));
    HcPtrs outs; memset(&outs, 0, sizeof outs); for (int z = 0; z < n; z++) outs.p[z] = ct_out[z];
    return hc_pack_run(c, c->ws_cts, cstride, n, max_ob, max_ob / norm, any_bias ? &bias : nullptr, 0, &outs);      // the root node writes ct_out itself
}
extern "C" int hc_conv_then_pack(hc_ctx *c, const uint64_t *ct_in, double ct_scale, const hc_ker *ker, double ker_scale,
                                 int max_ob, int norm, double out_scale, const uint64_t *bias, uint64_t *ct_out, double *scale_out) {
    HC_ENTER(c);
    if (!ct_in || !ker || !ct_out || ker->max_ob < max_ob) return hc_fail(c, HC_ERR_ARG, "hc_conv_then_pack: bad arguments");
    u64 cst[2]; double target;
    HC_TRY(hc_loopA_consts(c, ct_scale, ker_scale, max_ob, norm, out_scale, cst, &target));
    // conv.go:274 multiplies the scale by real_cnum; conv.go:541 then demands out_scale and level 0
    const double final_scale = target * (double)(max_ob / norm);
    if (final_scale != out_scale) return hc_fail(c, HC_ERR_STATE, "LV or scale after conv then pack, inconsistent");
    u64 *outs[1] = {(u64 *)ct_out};
    HC_TRY(hc_conv_batch_run(c, 1, hc_ptrs1((const u64 *)ct_in), hc_ptrs1(ker->d), hc_ptrs1((const u64 *)bias), bias != nullptr, outs, max_ob, norm, cst));
    if (scale_out) *scale_out = final_scale;
    return HC_OK;
}
extern "C" int hc_conv_then_pack_batch(hc_ctx *c, int n, const uint64_t *const *ct_in, double ct_scale, const hc_ker *const *ker, double ker_scale,
                                       int max_ob, int norm, double out_scale, const uint64_t *const *bias, uint64_t *const *ct_out, double *scale_out) {
    HC_ENTER(c);
    if (n < 1 || n > HC_MAXB) return hc_fail(c, HC_ERR_ARG, "hc_conv_then_pack_batch: n=%d outside 1..%d", n, HC_MAXB);
    if (!ct_in || !ker || !ct_out) return hc_fail(c, HC_ERR_ARG, "hc_conv_then_pack_batch: null");
    HcPtrs pin, pker, pbias; memset(&pin, 0, sizeof pin); memset(&pker, 0, sizeof pker); memset(&pbias, 0, sizeof pbias);
    u64 *outs[HC_MAXB]; bool any_bias = false;
    for (int z = 0; z < n; z++) {
        if (!ct_in[z] || !ker[z] || !ct_out[z] || ker[z]->max_ob < max_ob) return hc_fail(c, HC_ERR_ARG, "hc_conv_then_pack_batch: bad arguments for ciphertext %d", z);
        pin.p[z] = (const u64 *)ct_in[z]; pker.p[z] = ker[z]->d; outs[z] = (u64 *)ct_out[z];
        if (bias && bias[z]) { pbias.p[z] = (const u64 *)bias[z]; any_bias = true; }
    }
    u64 cst[2]; double target;
    HC_TRY(hc_loopA_consts(c, ct_scale, ker_scale, max_ob, norm, out_scale, cst, &target));
    const double final_scale = target * (double)(max_ob / norm);
    if (final_scale != out_scale) return hc_fail(c, HC_ERR_STATE, "LV or scale after conv then pack, inconsistent");
    HC_TRY(hc_conv_batch_run(c, n, pin, pker, pbias, any_bias, outs, max_ob, norm, cst));
    if (scale_out) *scale_out = final_scale;
    return HC_OK;
}

// ONE convolution sharded over G devices (BASELINE config 3: `conv 7 3`, B = 256 over 8 GPUs). conv.go:525-531's B products are
// independent and conv.go:286-297's tree pairs (i, i + step), so with the output channels dealt i mod G every tree level with
// step >= G is local to a device; only the last log2 G levels need the G partial ciphertexts (1 MiB each) in one place.
// ctxs[g] lives on device g (any devices; all on one device works too and is how the one-GPU boxes exercise this path), holds the
// same keys, a replica of the input ciphertext ct_in[g] and the kernel plaintexts ker[g] (all B of them; device g uses channels
// g, g + G, ...). Device g: loop A on its channels + the strided local tree, on its own stream. Device 0: waits on G events, pulls
// the partials with hipMemcpyPeerAsync (xGMI between GPUs of a node) on ITS stream, runs the last log2 G levels and the bias.
// No host synchronisation anywhere; the call returns with everything queued. Same arithmetic per node => same bits as one device.
extern "C" int hc_conv_then_pack_sharded(hc_ctx *const *ctxs, int G, const uint64_t *const *ct_in, double ct_scale, const hc_ker *const *ker, double ker_scale,
                                         int max_ob, double out_scale, const uint64_t *bias, uint64_t *ct_out, double *scale_out) {
    if (!ctxs || G < 1 || !ctxs[0]) return HC_ERR_ARG;
    hc_ctx *c0 = ctxs[0];
    HC_ENTER(c0);
    if (G > HC_MAXB || (G & (G - 1)) || !ct_in || !ker || !ct_out) return hc_fail(c0, HC_ERR_ARG, "hc_conv_then_pack_sharded: G=%d must be a power of two <= %d, pointers non-null", G, HC_MAXB);
    if (max_ob < G || max_ob % G) return hc_fail(c0, HC_ERR_ARG, "hc_conv_then_pack_sharded: max_ob=%d must be a multiple of G=%d", max_ob, G);
    int log2g = 0; while ((1 << log2g) < G) log2g++;
    const int nloc = max_ob / G;
    u64 cst[2]; double target;
    HC_TRY(hc_loopA_consts(c0, ct_scale, ker_scale, max_ob, 1, out_scale, cst, &target));
    if (target * (double)max_ob != out_scale) return hc_fail(c0, HC_ERR_STATE, "LV or scale after conv then pack, inconsistent");
    for (int g = 0; g < G; g++) {
        hc_ctx *c = ctxs[g];
        if (!c || !ct_in[g] || !ker[g] || ker[g]->max_ob < max_ob) return hc_fail(c0, HC_ERR_ARG, "hc_conv_then_pack_sharded: bad arguments for device slot %d", g);
        HC_ENTER(c);
        if (c->nq != c0->nq || c->mods[0].m.q != c0->mods[0].m.q || c->mods[1].m.q != c0->mods[1].m.q) return hc_fail(c0, HC_ERR_ARG, "hc_conv_then_pack_sharded: contexts differ in their moduli");
        if (!c->ev_shard) HC_HIP(c, hipEventCreateWithFlags(&c->ev_shard, hipEventDisableTiming));
        HC_TRY(hc_prepare_ctc(c, hc_ptrs1((const u64 *)ct_in[g]), 1, cst));
        HC_TRY(hc_ensure_cts(c, (size_t)nloc * 2));
        HC_TRY(hc_loopA_run_set(c, hc_ptrs1(ker[g]->d), 1, g, G, nloc, c->ws_cts, 0, true));     // channels g, g + G, ...: slot m = channel g + G m
        HC_TRY(hc_pack_run(c, c->ws_cts, 0, 1, nloc, nloc, nullptr, log2g));                      // levels with step >= G
        HC_HIP(c, hipEventRecord(c->ev_shard, c->stream));
        if (c != c0 && c->device != c0->device && c0->peer_access) {                                // direct xGMI copies when the devices can
            int can = 0;
            HC_HIP(c0, hipDeviceCanAccessPeer(&can, c0->device, c->device));
            if (can) {
                HC_HIP(c0, hipSetDevice(c0->device));
                const hipError_t pe = hipDeviceEnablePeerAccess(c->device, 0);
                if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) { if (pe != hipSuccess) (void)hipGetLastError(); c0->peer_enabled |= 1u << (unsigned)(c->device & 31); }
                else {      // recoverable: hipMemcpyPeerAsync stages the copy itself without peer access (same bits, slower); say so once per device pair
                    (void)hipGetLastError();
                    if (!(c0->peer_warned & (1u << (unsigned)(c->device & 31)))) { fprintf(stderr, "libhconv: hipDeviceEnablePeerAccess(device %d -> %d): %s; peer copies will be staged\n", c0->device, c->device, hipGetErrorString(pe)); c0->peer_warned |= 1u << (unsigned)(c->device & 31); }
                }
            }     // else: hipMemcpyPeerAsync stages through the host (slower, same bits)
        }
    }
    HC_ENTER(c0);
    if (c0->ws_gather_rows < (size_t)G * 2) {
        HC_HIP(c0, hipStreamSynchronize(c0->stream));
        if (c0->ws_gather) HC_HIP(c0, hcx_free(c0, c0->ws_gather));
        c0->ws_gather = nullptr; c0->ws_gather_rows = 0;
        HC_HIP(c0, hcx_malloc(c0, (void **)&c0->ws_gather, (size_t)G * 2 * HC_N * sizeof(u64)));
        c0->ws_gather_rows = (size_t)G * 2;
    }
    for (int g = 0; g < G; g++) {
        hc_ctx *c = ctxs[g];
        if (c != c0) HC_HIP(c0, hipStreamWaitEvent(c0->stream, c->ev_shard, 0));
        HC_HIP(c0, hipMemcpyPeerAsync(c0->ws_gather + (size_t)g * 2 * HC_N, c0->device, c->ws_cts, c->device, 2 * HC_N * sizeof(u64), c0->stream));
    }
    if (!c0->ev_fork) HC_HIP(c0, hipEventCreate(&c0->ev_fork));
    HC_HIP(c0, hipEventRecord(c0->ev_fork, c0->stream));                                            // the partials have been collected:
    for (int g = 1; g < G; g++) if (ctxs[g] != c0) { HC_HIP(c0, hipSetDevice(ctxs[g]->device)); HC_HIP(c0, hipStreamWaitEvent(ctxs[g]->stream, c0->ev_fork, 0)); }   // ... device g may reuse its workspace
    HC_ENTER(c0);
    const HcPtrs bp = hc_ptrs1((const u64 *)bias);
    HC_TRY(hc_pack_run(c0, c0->ws_gather, 0, 1, G, G, bias ? &bp : nullptr, 0));                    // the last log2 G levels + eval.go:258's bias
    HC_HIP(c0, hipMemcpyAsync(ct_out, c0->ws_gather, 2 * HC_N * sizeof(u64), hipMemcpyDeviceToDevice, c0->stream));
    if (scale_out) *scale_out = out_scale;
    return HC_OK;
}

// ------------------------------------------------------------------ slot encoder (ckks.Encoder.EncodeNTT), BL baseline's plaintexts
static int hc_enc_tables(hc_ctx *c) {
    if (c->enc_roots) return HC_OK;
    const int m = 2 * HC_N, slots = HC_N / 2;
    std::vector<HcCplx> roots((size_t)m + 1); std::vector<int> rg((size_t)slots);
    for (int i = 0; i < m; i++) { const double angle = 2 * 3.141592653589793 * (double)i / (double)m; roots[(size_t)i].re = hc_gomath::go_cos(angle); roots[(size_t)i].im = hc_gomath::go_sin(angle); }   // ckks.NewEncoder, with Go's own math.Cos / math.Sin
    roots[(size_t)m] = roots[0];
    int g = 1; for (int i = 0; i < slots; i++) { rg[(size_t)i] = g; g = (int)(((long)g * 5) % m); }
    HcScratch S(c);
    decltype(c->enc_roots) d_roots = nullptr; decltype(c->enc_rot_group) d_rg = nullptr;      // the context sees the tables only once both uploads succeeded
    HC_HIP(c, S.alloc(&d_roots, roots.size() * sizeof(HcCplx))); HC_HIP(c, S.alloc(&d_rg, rg.size() * sizeof(int)));
    HC_HIP(c, hcx_h2d(c, d_roots, roots.data(), roots.size() * sizeof(HcCplx)));
    HC_HIP(c, hcx_h2d(c, d_rg, rg.data(), rg.size() * sizeof(int)));
    S.keep(d_roots); S.keep(d_rg);
    c->enc_roots = d_roots; c->enc_rot_group = d_rg;
    return HC_OK;
}
// values: DEVICE [count][N/2] complex128 (re, im); overwritten (the transform runs in place). out: DEVICE [count][level+1][N]
extern "C" int hc_encode_slots(hc_ctx *c, double *values, int count, int level, double scale, int to_ntt, uint64_t *out) {
    HC_ENTER(c);
    if (!values || !out || count < 1 || count > 65535 || level < 0 || level >= c->nq) return hc_fail(c, HC_ERR_ARG, "hc_encode_slots: bad arguments");
    HC_TRY(hc_enc_tables(c));
    HcSlotEnc E; E.roots = c->enc_roots; E.rot_group = c->enc_rot_group;
    HcCplx *v = (HcCplx *)values;
    HC_TRY(hc_launch(c, "sfft_inv_a", hc_k_sfft_inv_a, dim3(16, (unsigned)count), (const HcCplx *)v, v, E));
    HC_TRY(hc_launch(c, "sfft_inv_b", hc_k_sfft_inv_b, dim3(16, (unsigned)count), (const HcCplx *)v, v, E));
    HC_TRY(hc_launch(c, "slots_round", hc_k_slots_round, dim3(64, (unsigned)count), (const HcCplx *)v, (u64 *)out, (const HcMod *)c->d_mods, level + 1, scale));
    if (to_ntt) { c->hoist_cx = nullptr; HC_TRY(hc_ntt_mm(c, (const u64 *)out, (u64 *)out, level + 1, level + 1, 0, 0, count, (size_t)(level + 1) * HC_N, (size_t)(level + 1) * HC_N)); }
    return HC_OK;
}
// out[2][level+1][N] = sum over t < ntaps of cts[t] (ciphertext [2][level+1][N]) x pts[t] (plaintext [level+1][N], NTT domain): the
// MulNew / Add chain of conv.go:167-172 in one launch. cts: HOST array of ntaps device pointers; pts: device, [ntaps][level+1][N]
extern "C" int hc_lv_mul_sum(hc_ctx *c, int level, const uint64_t *const *cts, const uint64_t *pts, int ntaps, uint64_t *out) {
    HC_ENTER(c);
    if (!cts || !pts || !out || ntaps < 1 || ntaps > HC_MAXTAPS || level < 0 || level >= c->nq) return hc_fail(c, HC_ERR_ARG, "hc_lv_mul_sum: bad arguments (1 <= ntaps <= %d)", HC_MAXTAPS);
    if (c->pack32 == 2) return hc_fail(c, HC_ERR_UNSUPPORTED, "hc_lv_mul_sum: the baseline's tap sum reads 8-byte rows only (option pack32 = 2 is for the bootstrapping chain's contexts)");
    HcTapPtrs T; memset(&T, 0, sizeof T);
    for (int t = 0; t < ntaps; t++) { if (!cts[t]) return hc_fail(c, HC_ERR_ARG, "hc_lv_mul_sum: null ciphertext %d", t); T.ct[t] = (const u64 *)cts[t]; }
    return hc_launch(c, "lv_mul_sum", hc_k_lv_mul_sum, dim3(64, (unsigned)(level + 1), 2), T, (const u64 *)pts, ntaps, level + 1, (u64 *)out, (const HcMod *)c->d_mods);
}
extern "C" int hc_bl_post_ker_slots(hc_ctx *c, const double *max_ker_rs, int in_wid, int ker_wid, int pad, int max_batch, int rot, double *values_out) {
    HC_ENTER(c);
    if (!max_ker_rs || !values_out || in_wid < 1 || ker_wid < 1 || max_batch < 1 || (long)max_batch * in_wid * in_wid > HC_N / 2) return hc_fail(c, HC_ERR_ARG, "hc_bl_post_ker_slots: bad arguments");
    return hc_launch(c, "bl_post_ker", hc_k_bl_post_ker, dim3(128, (unsigned)(ker_wid * ker_wid)), max_ker_rs, (HcCplx *)values_out, in_wid, ker_wid, pad, max_batch, rot);
}

// ------------------------------------------------------------------ options / timing
extern "C" int hc_row_is32(hc_ctx *c, int mod) {
    if (!c || mod < 0 || mod >= c->nq + c->np) return 0;
    return c->mods[(size_t)mod].m.row32 ? 1 : 0;
}
extern "C" int hc_set_option(hc_ctx *c, const char *name, long value) {
    if (!c || !name) return HC_ERR_ARG;
    if (!strcmp(name, "chunk_nodes")) { if (value < 1) return hc_fail(c, HC_ERR_ARG, "chunk_nodes must be >= 1"); c->chunk_nodes = value; return HC_OK; }
    if (!strcmp(name, "small_levels")) { if (value < 0) return hc_fail(c, HC_ERR_ARG, "small_levels must be >= 0"); c->small_levels = value; return HC_OK; }
    if (!strcmp(name, "peer_access")) { c->peer_access = value ? 1 : 0; return HC_OK; }
    if (!strcmp(name, "profile")) { hc_prof_flush(c); c->profile = value ? 1 : 0; return HC_OK; }
    if (!strcmp(name, "small_mm_wgs")) { if (value < 0) return hc_fail(c, HC_ERR_ARG, "small_mm_wgs must be >= 0"); c->small_mm_wgs = value; return HC_OK; }      // same residues either way
    if (!strcmp(name, "rot_fuse")) { c->rot_fuse = value ? 1 : 0; return HC_OK; }                  // same residues either way: A/B only
    if (!strcmp(name, "small32")) { HC_ENTER(c); c->small32 = value ? 1 : 0; HC_HIP(c, hipStreamSynchronize(c->stream)); return hc_upload_rowmods(c); }      // results do not depend on it (both forms leave canonical residues): A/B only
    if (!strcmp(name, "async_alloc")) {     // 0: hipMalloc / hipFree; 1: non-blocking stream + this context's cache of blocks (hcx_malloc); 2: diagnostic (ROCm's stream-ordered allocator)
        if (value < 0 || value > 2) return hc_fail(c, HC_ERR_ARG, "async_alloc must be 0, 1 or 2");
        if ((int)value == c->async_alloc) return HC_OK;
        // the stream and the ownership of every block follow the mode: it can only change while the context owns nothing but its tables (right after hc_ctx_create)
        if (c->allocs_live || !c->evk.empty() || !c->swk.empty() || c->idx_pairs || c->ws_cts || c->ws_tmp || c->ws_mm || c->ws_ctc || !c->cache_blk.empty())
            return hc_fail(c, HC_ERR_STATE, "async_alloc: set it right after hc_ctx_create, before the context allocates");
        HC_ENTER(c); HC_HIP(c, hipStreamSynchronize(c->stream));
        hipStream_t ns = nullptr;
        HC_HIP(c, value ? hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) : hipStreamCreate(&ns));
        HC_HIP(c, hipStreamDestroy(c->stream));
        c->stream = ns; c->async_alloc = (int)value;
        return HC_OK;
    }
    if (!strcmp(name, "pack32")) {          // 0 / 1 / 2 (include/hconv.h "4-byte rows"). Switching keys are stored per the setting in force when they are loaded or generated: 0 <-> 1, 2 only on a context without keys
        if (value < 0 || value > 2) return hc_fail(c, HC_ERR_ARG, "pack32 must be 0, 1 or 2");
        if (((value == 0) != (c->pack32 == 0)) && !c->swk.empty()) return hc_fail(c, HC_ERR_STATE, "pack32: switching keys are already stored in the other form");
        HC_ENTER(c); HC_HIP(c, hipStreamSynchronize(c->stream));
        c->pack32 = (int)value; c->hoist_cx = nullptr;
        std::vector<HcMod> hm; for (auto &mh : c->mods) { mh.m.row32 = (value == 2 && mh.m.q < (1ull << 31)) ? 1 : 0; hm.push_back(mh.m); }
        // limbs 0 and 1 are what the convolution's entry points, hc_div_round_last_n's level-1 branch and hc_swk_generate's sk rows read as 8-byte rows
        if (value == 2 && c->mods.size() > 1 && (hm[0].row32 || hm[1].row32)) {
            for (size_t l = 0; l < 2; l++) { c->mods[l].m.row32 = 0; hm[l].row32 = 0; }
        }
        HC_HIP(c, hcx_h2d(c, c->d_mods, hm.data(), hm.size() * sizeof(HcMod)));
        return HC_OK;
    }
    return hc_fail(c, HC_ERR_ARG, "unknown option %s", name);
}
extern "C" int hc_timer_start(hc_ctx *c) { HC_ENTER(c); HC_HIP(c, hipEventRecord(c->t0, c->stream)); return HC_OK; }
extern "C" int hc_timer_stop(hc_ctx *c, float *ms) {
    HC_ENTER(c); HC_HIP(c, hipEventRecord(c->t1, c->stream)); HC_HIP(c, hipEventSynchronize(c->t1));
    if (ms) HC_HIP(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return HC_OK;
}
extern "C" int hc_profile_get(hc_ctx *c, const char *name, double *total_ms, long *launches) {
    HC_ENTER(c); HC_TRY(hc_prof_flush(c));
    if (!name) { c->prof_acc.clear(); return HC_OK; }
    auto it = c->prof_acc.find(name);
    if (total_ms) *total_ms = it == c->prof_acc.end() ? 0 : it->second.first;
    if (launches) *launches = it == c->prof_acc.end() ? 0 : it->second.second;
    return HC_OK;
}
extern "C" int hc_profile_names(hc_ctx *c, char *buf, size_t buflen) {
    HC_ENTER(c); HC_TRY(hc_prof_flush(c));
    std::string s;
    for (auto &kv : c->prof_acc) { if (!s.empty()) s += ","; s += kv.first; }
    if (!buf || buflen < s.size() + 1) return hc_fail(c, HC_ERR_ARG, "hc_profile_names: buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return HC_OK;
}
