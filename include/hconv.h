/*
 * hconv.h — C ABI of libhconv.so: the MI355X (gfx950) engine behind the reference's `conv` hot path.
 *
 * The reference (github.com/dwkim606/optimal_conv, pure Go) has no FFI layer; the seam this library plugs into
 * is the `ckks.Evaluator` interface value every hot-path function receives:
 *     conv.go:522  conv_then_pack(params, pack_evaluator ckks.Evaluator, ...)
 *     conv.go:266  pack_ctxts(pack_eval ckks.Evaluator, ...)
 *     eval.go:224  evalConv_BN(cont *context, ...)      (uses cont.pack_evaluator / cont.evaluator, main.go:39-40)
 * A Go `gpuEvaluator` implements the methods those functions call and forwards each to one
 * entry point below over cgo. All signatures are plain C: opaque handles, raw pointers, sizes; no torch types.
 *
 * Conventions
 *  - Every function returns 0 on success, non-zero on error (hc_last_error gives the text). Nothing throws or
 *    aborts across the ABI; the Go shim turns non-zero into panic() to keep the reference's behaviour.
 *  - "dptr" arguments are DEVICE pointers obtained from hc_malloc. A "row" is N = 2^logN uint64 residues of one
 *    RNS limb, contiguous; a level-l polynomial is (l+1) consecutive rows; a ciphertext is [poly][limb][N].
 *    This is Lattigo's Poly.Coeffs[limb][N] layout made contiguous.
 *  - Moduli are addressed by index: 0..nq-1 = the Q chain (level order), nq..nq+np-1 = special primes P.
 *  - Values cross the ABI in Lattigo's public representation: ciphertext/plaintext rows are canonical residues
 *    in [0,q) in the NTT domain (bit-reversed order, psi chosen as ring.genNTTParams does); switching keys are
 *    handed over exactly as stored in rlwe.SwitchingKey.Value[d][k].Coeffs[limb] (NTT + Montgomery form).
 *  - The library never retains host pointers after a call returns (cgo rule); long-lived data is loaded through
 *    the *_load calls. Calls on one hc_ctx must be serialised by the caller; each call sets the device.
 *  - All work is queued on the context's own HIP stream; hc_sync waits for it. hc_download syncs implicitly.
 */
#ifndef HCONV_H
#define HCONV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hc_ctx hc_ctx;
typedef struct hc_ker hc_ker; /* device-resident kernel plaintexts pl_ker[0..max_ob) (conv.go:510-515) */

#define HC_OK 0
#define HC_ERR_ARG 1
#define HC_ERR_HIP 2
#define HC_ERR_STATE 3
#define HC_ERR_UNSUPPORTED 4

/* ---- context: replaces ckks.NewEvaluator(params, evk) for the pack evaluator (conv.go:258, main.go:446-461) ----
 * q[0..nq): ciphertext moduli in level order; p[0..np): special primes, np <= 5 (HC_ERR_UNSUPPORTED beyond: the reference's parameter sets use 1, 2 or 5). NTT tables are derived as
 * ring.genNTTParams does (psi = g^((q-1)/2N), g from ring.primitiveRoot). device = HIP device ordinal. */
int hc_ctx_create(hc_ctx **out, int logN, const uint64_t *q, int nq, const uint64_t *p, int np, int device);
void hc_ctx_destroy(hc_ctx *ctx);
const char *hc_last_error(const hc_ctx *ctx); /* ctx may be NULL: error of the last failed hc_ctx_create */
int hc_version(void);

/* ---- device memory ---- */
int hc_malloc(hc_ctx *ctx, size_t bytes, void **dptr);
int hc_free(hc_ctx *ctx, void *dptr);   /* into the context that allocated it; does not wait for the stream (blocks are recycled in stream order) */
int hc_upload(hc_ctx *ctx, void *dst_dptr, const void *src_host, size_t bytes);
int hc_download(hc_ctx *ctx, void *dst_host, const void *src_dptr, size_t bytes);
int hc_copy(hc_ctx *ctx, void *dst_dptr, const void *src_dptr, size_t bytes); /* device to device, on the stream */
int hc_sync(hc_ctx *ctx);
int hc_device_count(int *n);          /* HIP devices visible to the process */
/* device-to-device copy between the devices of two contexts (hipMemcpyPeerAsync: xGMI between the GPUs of a node), queued on
 * dst_ctx's stream behind everything src_ctx has queued so far; no host synchronisation */
int hc_copy_peer(hc_ctx *dst_ctx, void *dst_dptr, hc_ctx *src_ctx, const void *src_dptr, size_t bytes);

/* ---- L0: one call per ring / evaluator primitive (rows are device pointers; `count` consecutive rows) ---- */
/* ring.NTTLvl / ring.InvNTTLvl on limb `mod` (encoder.ToNTT conv.go:514; inside Rescale and key switching) */
int hc_ntt(hc_ctx *ctx, int mod, const uint64_t *in, uint64_t *out, int count);
int hc_intt(hc_ctx *ctx, int mod, const uint64_t *in, uint64_t *out, int count);
/* evaluator.MulNew(ct, pt) per limb (conv.go:527, conv.go:288): out = a*b mod q (= MForm + MulCoeffsMontgomery) */
int hc_mul(hc_ctx *ctx, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count);
/* evaluator.Add / SubNew per limb (conv.go:289,290,292; eval.go:258) */
int hc_add(hc_ctx *ctx, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count);
int hc_sub(hc_ctx *ctx, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out, int count);
/* evaluator.MultByConst's coefficient loop for an integer constant c (inside SetScale, conv.go:528) */
int hc_mul_const(hc_ctx *ctx, int mod, const uint64_t *a, uint64_t c, uint64_t *out, int count);
/* the integer constant and scale factor MultByConst derives from a float64 (getConstAndScale + scaleUpExact) */
uint64_t hc_const_for(double constant, double q_level_as_f64, uint64_t q, double *scale_mult);
/* ring.DivRoundByLastModulusNTT, one drop: x = (level+1) rows, out = level rows (inside Rescale/SetScale) */
int hc_div_round_last(hc_ctx *ctx, int level, const uint64_t *x, uint64_t *out);
/* the same drop on both polynomials of a ciphertext in one set of launches (evaluator.Rescale; x0 / x1 need not be adjacent) */
int hc_div_round_last2(hc_ctx *ctx, int level, const uint64_t *x0, const uint64_t *x1, uint64_t *out0, uint64_t *out1);
/* ring.PermuteNTTWithIndexLvl with ring.PermuteNTTIndex(galEl) (inside RotateGal, conv.go:291) */
int hc_permute(hc_ctx *ctx, uint64_t galEl, const uint64_t *in, uint64_t *out, int count);

/* ---- image batches of the leveled evaluator ----
 * The images of a batch go through a layer with the same weights, masks, DFT matrices and switching keys (test.go:128 runs them one after
 * another). hc_set_batch(n, poly_stride, qp_stride) makes every leveled entry point below (hc_lv_*, hc_rotate_finish, hc_keyswitch*, hc_mod_down2,
 * hc_qp_*, hc_div_round_last / 2) act on n <= 8 independent ciphertexts in ONE set of launches: a device pointer designates the operand of image 0,
 * image z's copy lies z * poly_stride words further (polynomials at a level: rows Q_0..Q_level) or z * qp_stride words further (extended-basis pairs
 * [2][level+1+np][N]: acc of hc_keyswitch_qp, x of hc_mod_down2, the operands of hc_qp_op2 / hc_qp_permute2). EVERY polynomial operand is per image. An operand that is
 * ONE PLAINTEXT for all images (a mask, an encoded diagonal, the secret key of the test harness) is named as such by the entry point or operation: b of hc_lv_mul_plain /
 * hc_lv_mul_acc_plain, b0 of HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN (hc_lv_op2, hc_qp_op2), the plaintexts of hc_qp_mul_sum*, every constant vector - it is read once per launch;
 * switching keys likewise (one fetch of a key row serves all images and both key components). (Until round 5 hc_lv_mul always shared b and hc_lv_op2 inferred a plaintext
 * from b0 == b1: a per-image second operand gave wrong residues for images above 0 without an error.) Results are bit-identical to n separate calls.
 * n = 1 (default) restores single-ciphertext behaviour; the L0 one-row primitives above, hc_permute and the L1 convolution (which has its own
 * batch entry point) ignore the setting. A decomposition held by hc_keyswitch_decompose belongs to the batch it was taken under.
 * The setting is context STATE (calls on one hc_ctx are serialised by the caller): a binding must hold it in a scope that restores n = 1 on every way out - INTEGRATION.md 3d
 * (`Batched` with a deferred reset), `Context.batch()` in abi.py, `Boot::Batch` in the C++ host. Under n > 1 every entry point checks the strides against the footprint of its
 * operands at the call's level - poly_stride >= (level+1) N, qp_stride >= 2 (level+1+np) N where it takes extended-basis pairs - and fails with HC_ERR_ARG otherwise (images
 * that overlap would race inside one launch). */
int hc_set_batch(hc_ctx *ctx, int n, size_t poly_stride_words, size_t qp_stride_words);

/* ---- L0, leveled: a polynomial at `level` is (level+1) consecutive rows, row l modulo q_l (Lattigo's ring.Poly / ckks.Element
 * at that level). One call covers all limbs. These are what a general-level ckks.Evaluator binds for the convReLU chain
 * (eval.go:272-607): MulNew/MulRelin's tensor products, Add/Sub, MultByConst / MulByPow2, AddConst, and the bootstrapper's modUp. */
int hc_lv_ntt(hc_ctx *ctx, int level, const uint64_t *in, uint64_t *out);
int hc_lv_intt(hc_ctx *ctx, int level, const uint64_t *in, uint64_t *out);
int hc_lv_mul(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *b, uint64_t *out);
/* acc += a * b: one term of a linear transform's diagonal sum (MulNew by the encoded diagonal + Add, conv.go:168-171 and the
 * bootstrapper's matrices) in one pass */
int hc_lv_mul_acc(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *b, uint64_t *acc);
/* the same with b = ONE plaintext for every image of a batch (read once per coefficient); identical to the above at n = 1 */
int hc_lv_mul_plain(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *pt, uint64_t *out);
int hc_lv_mul_acc_plain(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *pt, uint64_t *acc);
int hc_lv_add(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *b, uint64_t *out);
int hc_lv_sub(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *b, uint64_t *out);
/* the pointwise operations above on BOTH polynomials of a ciphertext in one launch (out_k = a_k op b_k, k = 0, 1; the polynomials may live
 * in separate allocations; HC_LV_MUL_CONST takes `consts` and ignores b; HC_LV_MUL_ACC accumulates into out; HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN: b0 is ONE plaintext that
 * multiplies both polynomials of every image, b1 must be NULL or b0) */
enum { HC_LV_MUL = 0, HC_LV_ADD = 1, HC_LV_SUB = 2, HC_LV_MUL_CONST = 3, HC_LV_MUL_ACC = 7, HC_LV_MUL_PLAIN = 8, HC_LV_MUL_ACC_PLAIN = 9 };
int hc_lv_op2(hc_ctx *ctx, int op, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *out0, uint64_t *out1, const uint64_t *consts);
/* evaluator.permuteNTT's tail after the key switch (RotateNew, RotateHoisted, ConjugateNew): out0 = Permute_galEl(d0 + c0),
 * out1 = Permute_galEl(d1) over limbs 0..level in one launch; same residues as hc_lv_add + two hc_permute calls */
int hc_rotate_finish(hc_ctx *ctx, uint64_t galEl, int level, const uint64_t *d0, const uint64_t *d1, const uint64_t *c0, uint64_t *out0, uint64_t *out1);
/* ring.PermuteNTTWithIndexLvl on a polynomial at `level` (rows 0..level) and on an extended-basis pair [2][level+1+np][N] (the same residues as
 * hc_permute over those rows), for every image of the batch; in != out */
int hc_lv_permute(hc_ctx *ctx, uint64_t galEl, int level, const uint64_t *in, uint64_t *out);
int hc_qp_permute2(hc_ctx *ctx, uint64_t galEl, int level, const uint64_t *in, uint64_t *out);
/* RotateNew / ConjugateNew / one rotation of RotateHoisted as a single call: key switch of c1 with key `key_id` (the key of galEl), + c0,
 * permutation of both polynomials, with the addition and the permutation inside ModDown's last pass. hoisted != 0: reuse the decomposition
 * left by hc_keyswitch_decompose(level, c1). Same residues as hc_keyswitch (or hc_keyswitch_hoisted) + hc_rotate_finish. */
int hc_keyswitch_rotate(hc_ctx *ctx, uint64_t key_id, uint64_t galEl, int level, const uint64_t *c0, const uint64_t *c1, uint64_t *out0, uint64_t *out1, int hoisted);
/* the tensor step of evaluator.mulRelin (conv.go:476; EvaluatePoly): d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1; outputs may not alias inputs */
int hc_lv_mul_tensor(hc_ctx *ctx, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1,
                     uint64_t *d0, uint64_t *d1, uint64_t *d2);
/* consts_host: level+1 host integers, one per limb (reduced mod q_l by the callee) */
int hc_lv_mul_const(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *consts_host, uint64_t *out);
int hc_lv_add_const(hc_ctx *ctx, int level, const uint64_t *a, const uint64_t *consts_host, uint64_t *out);
/* evaluatePolyFromPowerBasis' leaf (EvaluatePoly / EvaluateCheby: a MultByConst of every needed power and an Add chain, then AddConst) as ONE launch:
 * out_k[l] = sum over t < nterms (<= 8) of consts[t][l] * a_t,k[l]  (+ addc[l] on k = 0), k = 0, 1, rows 0..level. a0, a1: HOST arrays of nterms device pointers (the two
 * polynomials of each ciphertext); consts: HOST [nterms][level+1] integers, addc: HOST [level+1] or NULL (reduced by the callee). Exact modular sums: the same residues
 * as hc_lv_op2(HC_LV_MUL_CONST) + hc_lv_op2(HC_LV_ADD) per term + hc_lv_add_const. Outputs may alias inputs element-wise (out_k == a_t,k). */
int hc_lv_lincomb2(hc_ctx *ctx, int level, int nterms, const uint64_t *const *a0, const uint64_t *const *a1, const uint64_t *consts, const uint64_t *addc, uint64_t *out0, uint64_t *out1);
/* test_run: ckks.(*Bootstrapper).modUp for one polynomial: in_q0 = NTT row mod q_0; out = level+1 NTT rows of the centred lift */
int hc_lv_mod_raise(hc_ctx *ctx, int level, const uint64_t *in_q0, uint64_t *out);

/* rlwe.SwitchingKey for galEl, digit 0, the rows level-0 key switching reads: limb Q0 and the single P limb of
 * Value[0][0] (b) and Value[0][1] (a), HOST pointers, stored form. Replaces GenRotationKeys' output being handed
 * to NewEvaluator (conv.go:258). */
int hc_evk_load(hc_ctx *ctx, uint64_t galEl, const uint64_t *b_q, const uint64_t *a_q, const uint64_t *b_p,
                const uint64_t *a_p);
/* rlwe.KeySwitcher.SwitchKeysInPlace at level 0 (c1 -> d0,d1) and evaluator.RotateGal at level 0 (conv.go:291), for Galois
 * elements that permute inside 4096-coefficient tiles (2^j+1, j >= 5: every pack tree up to max_cnum 4096).
 * in/out may alias for hc_rotate_gal_l0 (conv.go:291 rotates in place). */
int hc_keyswitch_l0(hc_ctx *ctx, uint64_t galEl, const uint64_t *c1, uint64_t *d0, uint64_t *d1);
int hc_rotate_gal_l0(hc_ctx *ctx, uint64_t galEl, const uint64_t *c0, const uint64_t *c1, uint64_t *o0, uint64_t *o1);

/* General hybrid key switch, any level, alpha = np special primes, beta = ceil((level+1)/np) digits:
 * rlwe.KeySwitcher.SwitchKeysInPlace for an NTT-domain input (what RotateNew / Relinearize call outside the conv
 * path: BL baseline eval.go:123 at level 1 with two P primes; the bootstrapping chain with five).
 * hc_swk_load: HOST rows [beta][2][level+1+np][N] = rlwe.SwitchingKey.Value[d][k].Coeffs restricted to the Q limbs
 * 0..level followed by the np P limbs, stored form. hc_keyswitch: cx = (level+1) device rows (NTT); d0, d1 =
 * (level+1) device rows each, canonical. One launch per step covers all limbs (multi-modulus transforms, one basis extension
 * into every target limb, one accumulation of both key components): about 32 launches at level 27. */
int hc_swk_load(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *rows_host);
int hc_keyswitch(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *cx, uint64_t *d0, uint64_t *d1);
/* evaluator.Relinearize / MulRelin's tail as one call: out_k = a_k + (hc_keyswitch of cx)_k, k = 0, 1 (a = the degree-0 and degree-1 parts of the tensor product, cx = its
 * degree-2 part); the addition rides in ModDown's last pass. The same residues as hc_keyswitch + hc_lv_op2(HC_LV_ADD). out_k may be a_k. */
int hc_keyswitch_add(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *cx, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1);
/* hc_keyswitch_add followed by ONE hc_div_round_last2 (evaluator.MulRelin + Rescale's first drop), as one call: out_k = Rescale(a_k + (key switch of cx)_k), level >= 2,
 * out at level - 1. Same residues as the two calls; ModDown and the rescale share one forward transform per limb. */
int hc_keyswitch_add_rescale(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *cx, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1);
/* HARNESS ONLY - not part of what a Lattigo host binds (it owns its keys and hands them over with hc_swk_load): rlwe.KeyGenerator.GenSwitchingKey on the device for
 * the C++ test harness, restricted to the rows a level-`level` key switch reads. galEl odd: the rotation / conjugation key of galEl (s_out = sigma_{galEl^-1}(s));
 * galEl = 0: the relinearisation key (s^2 -> s). sk_ntt: DEVICE rows [nq + np][N] = NTT(s) modulo every modulus of the context. seed8: 8 x 32 bits keying ChaCha20;
 * uniform rows and the per-digit error (sigma 3.2, |e| <= 19) are functions of (seed, key_id, digit, limb, coefficient). The reference's keys are crypto/rand draws:
 * nothing to reproduce but the RLWE relation b + a s_out - [own limbs] P s_in = e, which tests/ check. The key is stored as hc_swk_load would store it. */
int hc_swk_generate(hc_ctx *ctx, uint64_t key_id, int level, uint64_t galEl, const uint64_t *sk_ntt, const uint32_t *seed8);
/* Hoisted form (evaluator.RotateHoisted, conv.go:131; Lattigo's linear transforms): hc_keyswitch_decompose computes the digit
 * decomposition of cx once and keeps it in the context; each hc_keyswitch_hoisted(key, level, cx, ...) then only does the inner
 * product with its key and the ModDown. Bit-identical to hc_keyswitch. The decomposition is valid until the next hc_keyswitch /
 * hc_keyswitch_decompose / hc_div_round_last on this context. */
int hc_keyswitch_decompose(hc_ctx *ctx, int level, const uint64_t *cx);
/* The key switch in two halves, and arithmetic in the extended basis QP (rows Q_0..Q_level, then P_0..P_(np-1)) between them: what Lattigo's
 * MultiplyByDiagMatrixBSGS (the linear transforms of CoeffsToSlots / SlotsToCoeffs; test_run @52a580) is made of.
 *  hc_keyswitch_qp  rlwe.(*KeySwitcher).SwitchKeysInPlaceNoModDown (@4fe660; hoisted = 0) / KeyswitchHoistedNoModDown (@4ff060; hoisted != 0: uses
 *                   the decomposition hc_keyswitch_decompose(level, cx) left): acc[2][level+1+np][N], canonical residues, NTT domain.
 *  hc_mod_down2     ring.(*FastBasisExtender).ModDownSplitNTTPQ (@4e4c40) on the two polynomials x[2][level+1+np][N] -> out0, out1 [level+1][N].
 *                   hc_keyswitch == hc_keyswitch_qp followed by hc_mod_down2, bit for bit. A decomposition held by hc_keyswitch_decompose
 *                   survives hc_mod_down2 at the SAME level only (another level drops it); hc_keyswitch_qp(hoisted = 0) always re-decomposes and drops it.
 *  hc_qp_op2        out_k = a_k (op) b_k for k = 0, 1 over all level+1+np rows; op = HC_LV_MUL, HC_LV_ADD, HC_LV_MUL_ACC (out_k += a_k * b_k) or their _PLAIN forms (b0 = one plaintext);
 *                   a plaintext operand (an encoded diagonal) is ONE polynomial shared by both components and by every image of a batch: say so with HC_LV_MUL_PLAIN /
 *                   HC_LV_MUL_ACC_PLAIN (b1 NULL or b0). HC_LV_MUL / HC_LV_MUL_ACC with b0 == b1 inside an image batch (hc_set_batch n > 1) is refused with HC_ERR_ARG since
 *                   hc_version() 2: version 1 inferred "shared plaintext" from it, and a per-image operand would now be read past the plaintext's allocation.
 *                   (hc_permute works on any rows, so it permutes QP polynomials as they are.) */
int hc_keyswitch_qp(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *cx, uint64_t *acc, int hoisted);
/* one rotation of MultiplyByDiagMatrixBSGS as a single call (rotateHoistedNoModDown for a baby step: pc0 = P * c0; the giant step's SwitchKeysInPlaceNoModDown +
 * permutation: pc0 = NULL): out[2][level+1+np][N] (+)= Permute_galEl( hc_keyswitch_qp(cx) + (pc0 on the Q rows of the first component) ); accumulate != 0 adds to
 * what out holds. The same residues as hc_keyswitch_qp + hc_lv_add + hc_qp_permute2 (+ hc_qp_op2 HC_LV_ADD); out must not be the scratch of another call. */
int hc_keyswitch_qp_rotate(hc_ctx *ctx, uint64_t key_id, uint64_t galEl, int level, const uint64_t *pc0, const uint64_t *cx, uint64_t *out, int hoisted, int accumulate);
/* all baby steps of a linear transform in one call: nrot hoisted rotations (key_ids[r], galEls[r]) of the decomposition hc_keyswitch_decompose(level, cx) holds, outs[r] as
 * hc_keyswitch_qp_rotate(key_ids[r], galEls[r], level, pc0, cx, outs[r], 1, 0) leaves it. The inner products of several rotations share one pass over the digits.
 * key_ids, galEls, outs: HOST arrays. */
int hc_keyswitch_qp_rotate_many(hc_ctx *ctx, int nrot, const uint64_t *key_ids, const uint64_t *galEls, int level, const uint64_t *pc0, const uint64_t *cx, uint64_t *const *outs);
int hc_mod_down2(hc_ctx *ctx, int level, const uint64_t *x, uint64_t *out0, uint64_t *out1);
/* hc_mod_down2, + a_k, and ONE hc_div_round_last2 as one call (the end of a linear transform): out_k = Rescale(ModDown(x)_k + a_k) at level - 1 (level >= 2); a0, a1 both
 * NULL = no addend. Same residues as the three calls; row `level` of x is overwritten. */
int hc_mod_down2_add_rescale(hc_ctx *ctx, int level, uint64_t *x, const uint64_t *a0, const uint64_t *a1, uint64_t *out0, uint64_t *out1);
int hc_qp_op2(hc_ctx *ctx, int op, int level, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *out0, uint64_t *out1);
int hc_keyswitch_hoisted(hc_ctx *ctx, uint64_t key_id, int level, const uint64_t *cx, uint64_t *d0, uint64_t *d1);
/* the diagonal sum of one giant step of MultiplyByDiagMatrixBSGS in one launch: out[2][level+1+np][N] (+)= sum over t < nterms (<= 64) of a[t] (*) pt[t], a[t] =
 * extended-basis pairs (the hoisted rotations), pt[t] = plaintexts [level+1+np][N] (the encoded diagonals); a, pt: HOST arrays of device pointers; accumulate != 0
 * adds to what out holds. The same residues as nterms hc_qp_op2 calls (HC_LV_MUL, then HC_LV_MUL_ACC); out must not be one of the a[t]. */
int hc_qp_mul_sum(hc_ctx *ctx, int level, int nterms, const uint64_t *const *a, const uint64_t *const *pt, uint64_t *out, int accumulate);
/* two giant steps in one pass over the rotations: out_h (+)= sum_t a[t] (*) pt_h[t], h = 0, 1 (pt0[t] or pt1[t] NULL: that giant step has no diagonal for baby step t).
 * Same residues as two hc_qp_mul_sum calls; the rotated ciphertexts are read once. a, pt0, pt1: HOST arrays of device pointers. */
int hc_qp_mul_sum2(hc_ctx *ctx, int level, int nterms, const uint64_t *const *a, const uint64_t *const *pt0, const uint64_t *const *pt1, uint64_t *out0, uint64_t *out1, int accumulate0, int accumulate1);
/* the same for ngiant <= 4 giant steps: out[h] (+)= sum_t a[t] (*) pt[h * nterms + t] (NULL entries: no diagonal), accumulate[h] per output. HOST arrays. */
int hc_qp_mul_sum_many(hc_ctx *ctx, int level, int nterms, int ngiant, const uint64_t *const *a, const uint64_t *const *pt, uint64_t *const *out, const int *accumulate);

/* ---- L1: the fused hot path ---- */
/* pl_ker as prep_Ker leaves it (conv.go:510-515): HOST array [max_ob][2][N], level 1, NTT domain. */
int hc_ker_load(hc_ctx *ctx, const uint64_t *pl_ker_host, int max_ob, hc_ker **out);
/* same from a DEVICE array (e.g. produced by hc_ntt); the input buffer is not retained */
int hc_ker_load_device(hc_ctx *ctx, const uint64_t *pl_ker_dptr, int max_ob, hc_ker **out);
/* prep_Ker itself (conv.go:487-518; pos = 0, trans = false) on the device: HOST float arrays as the reference's readTxt
 * returns them (ker_in HWIO flat of length ker_len = k^2*real_ib*real_ob, BN_a of length real_ob) -> reshape_ker, BN
 * scaling, max_bat embedding with stride norm, encode_ker_final, EncodeCoeffs at `scale` (level 1), ToNTT. */
int hc_prep_ker(hc_ctx *ctx, const double *ker_in, int ker_len, const double *bn_a, int in_wid, int ker_wid,
                int real_ib, int real_ob, int norm, double scale, hc_ker **out);
/* the plaintexts of a handle as Lattigo would hold them: HOST out [max_ob][2][N], canonical NTT residues */
int hc_ker_download(hc_ctx *ctx, const hc_ker *ker, uint64_t *host_out);
void hc_ker_free(hc_ctx *ctx, hc_ker *ker);
/* plain_idx (conv.go:241-261): NULL => derive idx[s] = NTT(X^(2^s)) on the device; else HOST array [logN][N] */
int hc_idx_load(hc_ctx *ctx, const uint64_t *idx_host);

/* conv_then_pack (conv.go:522-546) followed by eval.go:258's bias add when bias != NULL.
 *   ct_in : device, [2][2][N] level-1 ciphertext, Scale = ct_scale
 *   ker   : kernel plaintexts with Scale = ker_scale; max_ob, norm as in conv.go:522
 *   bias  : device row mod Q0 (pl_bn_b, level 0, NTT) or NULL
 *   ct_out: device, [2][N] level-0 ciphertext; *scale_out = its Scale
 * Fails with HC_ERR_STATE when the resulting level/scale would trip the reference's panic (conv.go:541-543). */
int hc_conv_then_pack(hc_ctx *ctx, const uint64_t *ct_in, double ct_scale, const hc_ker *ker, double ker_scale,
                      int max_ob, int norm, double out_scale, const uint64_t *bias, uint64_t *ct_out,
                      double *scale_out);
/* The same convolution on n <= 16 independent ciphertexts (the images of a batch going through one layer: test.go:76-370 runs
 * them one after another through evalConv_BN, eval.go:224-263) as ONE set of kernel launches: every launch covers all n
 * ciphertexts, so the switching keys, idx plaintexts and twiddles are fetched once per launch and the top levels of the pack
 * tree (1..16 nodes each) are n times wider. ct_in, ker, bias (or NULL, or NULL entries), ct_out: HOST arrays of n device
 * pointers / handles (entries of ker and bias may repeat); same shapes and scales for all; results are bit-identical to n
 * separate hc_conv_then_pack calls. */
int hc_conv_then_pack_batch(hc_ctx *ctx, int n, const uint64_t *const *ct_in, double ct_scale, const hc_ker *const *ker,
                            double ker_scale, int max_ob, int norm, double out_scale, const uint64_t *const *bias,
                            uint64_t *const *ct_out, double *scale_out);
/* ONE convolution sharded over G devices (BASELINE config `conv 7 3` over 8 GPUs): ctxs[g] = a context on device
 * g with the same moduli and switching keys; ct_in[g] = that device's replica of the level-1 input; ker[g] = that device's
 * handle to all max_ob kernel plaintexts (device g multiplies the channels g, g + G, ...). Each device runs loop A and the tree
 * levels with step >= G on its own stream; device 0 collects the G partial ciphertexts (1 MiB each) with peer copies ordered by
 * events -- no host synchronisation, no other exchange -- and finishes the last log2 G levels and the bias. bias and ct_out are
 * on device 0. Bit-identical to hc_conv_then_pack on one device. G a power of two <= 16 dividing max_ob; contexts may share a device. */
int hc_conv_then_pack_sharded(hc_ctx *const *ctxs, int G, const uint64_t *const *ct_in, double ct_scale, const hc_ker *const *ker,
                              double ker_scale, int max_ob, double out_scale, const uint64_t *bias, uint64_t *ct_out,
                              double *scale_out);
/* loop A only (conv.go:525-531): cts_out = device [max_ob][2][N]; and loop B only (pack_ctxts, conv.go:266-300),
 * in place on cts (result in slot 0). Exposed for parity tests and profiling. */
int hc_conv_mult_phase(hc_ctx *ctx, const uint64_t *ct_in, double ct_scale, const hc_ker *ker, double ker_scale,
                       int max_ob, int norm, double out_scale, uint64_t *cts_out);
int hc_pack_ctxts(hc_ctx *ctx, uint64_t *cts, int max_cnum, int real_cnum);
/* The same tree over `count` level-0 ciphertexts whose indices in the full pack are m << stride_log2 (m = slot):
 * what one GPU holds when conv.go:286-297's tree is sharded by i mod G (stride_log2 = log2 G: the levels with
 * step >= G), and, with stride_log2 = 0, the last log2 G levels over the G gathered partial results. bias (device
 * row mod Q0 or NULL) is added to the result as eval.go:258 does. */
int hc_pack_ctxts_strided(hc_ctx *ctx, uint64_t *cts, int count, int stride_log2, const uint64_t *bias);

/* ---- slot encoder: ckks.Encoder.Encode (+ ToNTT = EncodeNTT), full slots (conv.go:165-166, eval.go:102; the BL baseline's plaintexts) ----
 * values: DEVICE [count][N/2] complex128 as (re, im) pairs; OVERWRITTEN (Lattigo's special inverse FFT runs in place). scale as in
 * Encode; level: rows 0..level (moduli 0..level) are produced; to_ntt != 0: the rows are left in the NTT domain. out: DEVICE
 * [count][level+1][N]. IEEE fp64 without contraction in the reference's operand order: the same residues as the CPU encoder. */
int hc_encode_slots(hc_ctx *ctx, double *values, int count, int level, double scale, int to_ntt, uint64_t *out);
/* conv.go:167-172 in one launch: out[2][level+1][N] = sum over t < ntaps (<= 64) of ciphertext cts[t] ([2][level+1][N], device) x
 * plaintext pts[t] ([level+1][N], NTT domain, device [ntaps][level+1][N]); cts is a HOST array of device pointers. Exact modular
 * sums: the same residues as the reference's MulNew + Add chain. */
int hc_lv_mul_sum(hc_ctx *ctx, int level, const uint64_t *const *cts, const uint64_t *pts, int ntaps, uint64_t *out);
/* conv.go:150-164 on the device: the slot vectors `postKer` of all ker_wid^2 kernel taps of output rotation `rot`, from
 * max_ker_rs = reshape_ker_BL's [ker_wid][ker_wid][max_batch][max_batch] doubles (DEVICE); values_out: DEVICE [ker_wid^2][N/2][2] */
int hc_bl_post_ker_slots(hc_ctx *ctx, const double *max_ker_rs, int in_wid, int ker_wid, int pad, int max_batch, int rot,
                         double *values_out);

/* ---- 4-byte rows (round 5) ----
 * Lattigo stores every residue in a uint64. Eleven of the 28 Q limbs of ckks.DefaultBootstrapParams[6] / [7] are ~30-bit primes, and every kernel of the bootstrapping chain is
 * priced in bytes: option "pack32" lets rows of limbs below 2^31 be stored as N 4-byte words AT THE ROW'S ADDRESS (the row pitch stays N 8-byte words: no stride, size or
 * pointer arithmetic of the caller changes; the second half of such a row's slot is simply unused).
 *   1 (default): inside the library only - the seam between the two passes of every transform, the extended digits of a key switch, the switching keys. Invisible at this ABI.
 *   2: ALSO in every LEVELED operand a caller hands in or gets back - the polynomials of hc_lv_*, hc_rotate_finish, hc_lv_permute, hc_keyswitch*, hc_div_round_last / 2 (general
 *      level), the extended-basis pairs of hc_keyswitch_qp*, hc_mod_down2*, hc_qp_*, the plaintexts they multiply by, hc_encode_slots' output. hc_row_is32(ctx, mod) tells which
 *      limbs that concerns; a caller converts at its own boundary only (what it uploads into / downloads from such rows: the C++ host's Boot::put_rows / get_rows; ciphertexts
 *      enter and leave the chain at levels 0 / 1, whose limbs are large, so the hot path converts nothing). The L0 one-row primitives (hc_ntt ... hc_permute with an explicit
 *      modulus or row count), the level-0/1 convolution path and hc_swk_generate's secret-key rows keep 8-byte rows. Same residues, bit for bit, in every setting.
 *   0: off. Keys are stored per the setting in force when they are loaded: switch between 0 and 1 / 2 only on a context without keys (HC_ERR_STATE otherwise).
 * Setting 2 never applies to limbs 0 and 1 (the convolution's level-0 / 1 entry points and the secret-key rows read them as 8-byte rows whatever their size).
 * The library reads NO configuration from the environment (a cgo host would inherit its shell's): every switch is an hc_set_option; this repo's CLI translates
 * HCONV_PACK32 / HCONV_SMALL32 / HCONV_ROT_FUSE / HCONV_ASYNC_ALLOC into those calls (host/hconv_host.cpp applyEnvOptions). */
int hc_row_is32(hc_ctx *ctx, int mod);
/* ---- tuning / measurement ---- */
int hc_set_option(hc_ctx *ctx, const char *name, long value); /* "chunk_nodes" (jobs - channels / tree nodes summed over the batch - per kernel launch), "small_levels" (tree levels of at most
                                                                  this many nodes x ciphertexts run on the quarter-tile kernels: default 16, 0 = never), "profile" (per-kernel HIP-event totals),
                                                                  "peer_access" (hc_conv_then_pack_sharded: 0 = do not enable direct peer copies; default 1: enabled where hipDeviceCanAccessPeer allows),
                                                                  "pack32" (0 / 1 / 2: 4-byte rows, above), "rot_fuse" (default 1: hc_keyswitch_qp_rotate_many stores every rotation's result already permuted and with P c0 added from inside the inner
                                                                  product; 0: one pass per rotation over the accumulators - same residues, an A/B switch), "small32" (default 1: rows of a modulus below 2^31 take the 32-bit body of the batched
                                                                  transform kernels; 0: the 64-bit body for every row - the same residues either way, an A/B switch), "small_mm_wgs" (default 1024: a batched inverse pass / second forward pass of at most this many 16-row workgroups runs on quarter tiles - four residues per thread, four times the
                                                                  workgroups: such a launch costs one workgroup's latency; 0: never - the same residues either way), "async_alloc" (0, default: hipMalloc / hipFree; 1: a non-blocking stream and a per-context cache of blocks, so that hc_free never
                                                                  drains the device - for several contexts driven from several host threads; only right after hc_ctx_create, HC_ERR_STATE once the context owns memory) */
/* HIP-event timing on the context's stream */
int hc_timer_start(hc_ctx *ctx);
int hc_timer_stop(hc_ctx *ctx, float *ms);
/* per-kernel accumulated HIP-event time while option "profile" is 1; name==NULL resets */
int hc_profile_get(hc_ctx *ctx, const char *kernel_name, double *total_ms, long *launches);
int hc_profile_names(hc_ctx *ctx, char *buf, size_t buflen); /* comma separated */

#ifdef __cplusplus
}
#endif
#endif
