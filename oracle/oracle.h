/*
 * oracle.h — CPU restatement of the reference's homomorphic-convolution hot path.
 *
 * TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (optimal_conv_amd/, libhconv.so) never links, imports or calls it.
 *
 * What it restates:
 *   - the reference's own Go on the path: conv.go:184-237 (reshape_ker, encode_ker_final), conv.go:241-300
 *     (gen_idxNlogs, pack_ctxts), conv.go:487-546 (prep_Ker, conv_then_pack), eval.go:224-263 (evalConv_BN),
 *     main.go:1007-1042 (prep_Input), main.go:1057-1070 (post_process);
 *   - the arithmetic those call, which lives in an un-vendored dependency absent from /root/reference:
 *     github.com/dwkim606/test_lattigo v0.0.0-20220812213541-eb33b0555aaa (a fork of Lattigo v2.2.0; pin
 *     recovered from the build-info of /root/reference/test_run). Its published algorithms (ring.MRed/BRed,
 *     ring.NTT/InvNTT, ring.PermuteNTTIndex, rlwe.KeySwitcher.SwitchKeysInPlace, ring.FastBasisExtender.
 *     ModDownSplitNTTPQ, ring.DivRoundByLastModulusNTT, ckks.evaluator.{mulRelin,MultByConst,Rescale,SetScale,
 *     permuteNTT}, ckks.encoder.EncodeCoeffs) are restated here from the upstream v2.2.0 definitions and from
 *     the disassembly notes in SURVEY.md section 8(a)-R.
 *
 * Pinning: PINNED against the reference binary itself. oracle/pin/gotrace.c runs /root/reference/test_run
 * under ptrace with planted inputs and records SHA-256 digests of every intermediate ciphertext of
 * conv_then_pack; tests/test_oracle_pin.py replays the same inputs through this file and requires every
 * digest to match (tests/golden/ref_trace_conv_*.json).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct or_ctx or_ctx;

/* moduli are indexed 0..nq-1 for the Q chain, nq..nq+np-1 for the special primes P */
or_ctx *or_ctx_new(int logN, const uint64_t *q, int nq, const uint64_t *p, int np);
void or_ctx_free(or_ctx *);
int or_N(const or_ctx *);
uint64_t or_modulus(const or_ctx *, int mod);
/* pointers into the context's tables (length N each; Montgomery form, bit-reversed index as in ring.genNTTParams) */
const uint64_t *or_psi(const or_ctx *, int mod);
const uint64_t *or_psi_inv(const or_ctx *, int mod);
uint64_t or_primitive_root(uint64_t q);

/* ring ops on single limb rows of length N; in/out may alias */
void or_ntt(const or_ctx *, int mod, const uint64_t *in, uint64_t *out);       /* ring.NTT: canonical output */
void or_intt(const or_ctx *, int mod, const uint64_t *in, uint64_t *out);      /* ring.InvNTT: canonical output */
void or_mform(const or_ctx *, int mod, const uint64_t *in, uint64_t *out);     /* a * 2^64 mod q */
void or_mul_mont(const or_ctx *, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out); /* MRed(a,b) */
void or_mul(const or_ctx *, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out);      /* a*b mod q */
void or_mul_scalar(const or_ctx *, int mod, const uint64_t *a, uint64_t c, uint64_t *out);      /* a*c mod q */
void or_add(const or_ctx *, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out);
void or_sub(const or_ctx *, int mod, const uint64_t *a, const uint64_t *b, uint64_t *out);
void or_permute_index(int logN, uint64_t galEl, uint32_t *idx);                /* ring.PermuteNTTIndex */
void or_permute(int N, const uint32_t *idx, const uint64_t *in, uint64_t *out);/* out[i] = in[idx[i]]; no alias */

/* ckks.evaluator.MultByConst's integer constant for a float64 constant at modulus q (getConstAndScale +
 * scaleUpExact); *scale_mult receives the factor by which the ciphertext scale grows (1 or float64(q_level)) */
uint64_t or_const_for(double constant, double q_level_f, uint64_t q, double *scale_mult);
/* ckks.evaluator.Rescale's drop count: number of limbs dropped from `level` given scale and minScale */
int or_rescale_drops(const or_ctx *, int level, double scale, double min_scale, double *scale_out);
/* ring.DivRoundByLastModulusNTT for one drop: x has (level+1) limb rows (row stride N); writes rows 0..level-1 */
void or_div_round_last_ntt(const or_ctx *, int level, const uint64_t *x, uint64_t *out);

/* rlwe.KeySwitcher.SwitchKeysInPlace at level 0 with a single special prime (mod index nq):
 * c1: NTT row mod Q0; evk_*: the switching key's digit-0 rows in Lattigo's stored form (NTT + Montgomery);
 * d0,d1: canonical rows mod Q0 */
void or_keyswitch_l0(const or_ctx *, const uint64_t *c1, const uint64_t *evk_b_q, const uint64_t *evk_a_q,
                     const uint64_t *evk_b_p, const uint64_t *evk_a_p, uint64_t *d0, uint64_t *d1);
/* ---- general hybrid key switch (any level, alpha = np special primes, beta = ceil((level+1)/alpha) digits) ----
 * rlwe.KeySwitcher.SwitchKeysInPlace for an NTT-domain input: cx = (level+1) rows; evk = [beta][2][(level+1)+np][N]
 * rows of rlwe.SwitchingKey.Value[d][k] restricted to the Q limbs 0..level followed by the np P limbs, stored form
 * (NTT + Montgomery); d0,d1 = (level+1) canonical rows each. Restates DecomposeSingleNTT / ring.Decomposer.
 * DecomposeAndSplit (single-limb digits are copied, multi-limb digits go through reconstructRNS + multSum with the
 * fp64 overflow count), the Montgomery MAC over digits, and ring.FastBasisExtender.ModDownSplitNTTPQ with
 * ring.modUpExact over np primes. Used by the BL path (level 1, np = 2) and by everything bootstrapping needs. */
void or_keyswitch(const or_ctx *, int level, const uint64_t *cx, const uint64_t *evk, uint64_t *d0, uint64_t *d1);
/* the two halves of or_keyswitch (rlwe SwitchKeysInPlaceNoModDown / KeyswitchHoistedNoModDown, ring ModDownSplitNTTPQ):
 * acc = [2][level+1+np][N] in the basis Q_0..Q_level, P_0..P_(np-1); or_mod_down takes ONE such polynomial to [level+1][N] */
void or_keyswitch_qp(const or_ctx *, int level, const uint64_t *cx, const uint64_t *evk, uint64_t *acc);
/* or_keyswitch_qp in its own two steps (hoisting: one decomposition, several keys): digits = [beta][level+1+np][N] */
void or_keyswitch_decompose(const or_ctx *, int level, const uint64_t *cx, uint64_t *digits);
void or_keyswitch_mac(const or_ctx *, int level, const uint64_t *digits, const uint64_t *evk, uint64_t *acc);
void or_mod_down(const or_ctx *, int level, const uint64_t *x_qp, uint64_t *out);
/* the exact fast basis extension both steps use: residues x[0..n) (any representatives) modulo src[0..n) -> modulus t */
uint64_t or_basis_extend(const uint64_t *x, const uint64_t *src, int n, uint64_t t);

/* ring.modUpExact for one P prime -> one Q prime, per coefficient (exposes the fp64 overflow count) */
uint64_t or_modup_1p(uint64_t y, uint64_t p, uint64_t q);

/* ckks.evaluator.RotateGal at level 0: ct (c0,c1) -> (o0,o1) */
void or_rotate_gal_l0(const or_ctx *, const uint64_t *c0, const uint64_t *c1, const uint32_t *perm_idx,
                      const uint64_t *evk_b_q, const uint64_t *evk_a_q, const uint64_t *evk_b_p,
                      const uint64_t *evk_a_p, uint64_t *o0, uint64_t *o1);

/* Loop A body (conv.go:527-528): ct_in (2 polys x 2 limbs, layout [poly][limb][N]) times pl_ker (2 limbs),
 * then SetScale -> level 0 ciphertext out (2 polys x 1 limb, layout [poly][N]). cst[l] = integer constant. */
void or_mul_setscale(const or_ctx *, const uint64_t *ct_in, const uint64_t *pl_ker, const uint64_t cst[2],
                     uint64_t *ct_out);

/* conv.go:522-546 + eval.go:258 on residue arrays.
 *  ct_in  [2][2][N]; pl_ker [max_ob][2][N]; idx_pt [logN][N] (NTT(X^(2^s)) mod Q0, plain form);
 *  evk    [logN][4][N] ordered (b_q, a_q, b_p, a_p) for galEl 2^(s'+1)+1 at slot s' = j-1 (only the slots the
 *         tree touches are read); bias [N] or NULL; ct_out [2][N]. Returns the resulting scale. */
double or_conv_then_pack(const or_ctx *, const uint64_t *ct_in, double ct_scale, const uint64_t *pl_ker,
                         double ker_scale, const uint64_t *idx_pt, const uint64_t *evk, int max_ob, int norm,
                         double out_scale, const uint64_t *bias, uint64_t *ct_out);

/* ---- encoding / host-side layout (float64) ---- */
/* ckks.encoder.EncodeCoeffs -> scaleUpVecExact: out rows for `nmods` moduli mods[]; coefficient domain */
void or_encode_coeffs(const or_ctx *, const double *v, int n, double scale, const int *mods, int nmods, uint64_t *out);
void or_prep_input(const double *input, int raw_in_wid, int in_wid, int N, int norm, double *out);   /* main.go:1007 */
void or_reshape_ker(const double *ker_in, int len, int k_sz, int out_batch, double *ker_out);         /* conv.go:184 */
void or_encode_ker_final(const double *ker_rs, int row_len, int pos, int i, int in_wid, int in_batch, int ker_wid,
                         double *out);                                                              /* conv.go:206 */
/* conv.go:487-518 up to (not including) the encoder calls: float coefficient vectors, [max_bat][N] */
void or_prep_ker_coeffs(const double *ker_in, int ker_len, const double *bn_a, int N, int in_wid, int ker_wid,
                        int real_ib, int real_ob, int norm, double *out);
void or_post_process(const double *in_cfs, int len, int raw_in_wid, int in_wid, double *out);        /* main.go:1057 */
void or_bias_coeffs(const double *bn_b, int real_ob, int N, int in_wid, int norm, double *out);      /* eval.go:233-238 */

/* ---- harness-only crypto (own seeded PRNG; the reference's randomness is unseeded so only self-consistency
 *      is required: rotate/keyswitch must decrypt correctly) ---- */
void or_gen_sk(const or_ctx *, uint64_t seed, int h, int64_t *sk_coeffs);                  /* sparse ternary */
void or_sk_rows(const or_ctx *, const int64_t *sk_coeffs, int mod, uint64_t *out_ntt);     /* NTT, plain form */
/* switching key digit 0 at level 0 for galEl: rows (b_q, a_q, b_p, a_p) in stored form (NTT + Montgomery) */
void or_gen_galois_key_l0(const or_ctx *, const int64_t *sk_coeffs, uint64_t galEl, uint64_t seed, uint64_t *evk4);
/* general switching key for galEl at `level`: rows [beta][2][level+1+np][N] in stored form (see or_keyswitch);
 * galEl == 0 generates the relinearisation key (s_in = s^2, s_out = s) */
void or_gen_swk(const or_ctx *, const int64_t *sk_coeffs, uint64_t galEl, int level, uint64_t seed, uint64_t *rows);
/* sk-encryption of an encoded plaintext (coefficient domain rows for moduli 0..level) -> ct [2][level+1][N] NTT */
void or_encrypt(const or_ctx *, const int64_t *sk_coeffs, const uint64_t *pt_coeff_rows, int level, uint64_t seed,
                uint64_t *ct);
/* level-0 decrypt + DecodeCoeffs -> N doubles */
void or_decrypt_decode_l0(const or_ctx *, const int64_t *sk_coeffs, const uint64_t *ct, double scale, double *out);

/* counter-based splitmix64 residues (shared with oracle/pin/gotrace.c and the tests) */
void or_fill_seeded(uint64_t seed, uint64_t q, int n, uint64_t *out);

/* threads of the oracle's OpenMP row loops (default: the runtime's); see oracle.c */
void or_set_threads(int n);
#ifdef __cplusplus
}
#endif
#endif
