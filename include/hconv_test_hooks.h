/* hconv_test_hooks.h - entry points of libhconv.so that exist for THIS REPOSITORY'S TESTS ONLY. Nothing a Lattigo host binds is declared here (that is include/hconv.h);
 * a cgo shim never includes this file. The one hook lets the product host replay, bit for bit, a network the test oracle evaluated under its own (splitmix64) keys:
 * `conv --test-mode resnet ...` with HCONV_RESNET_REPLAY (host/hconv_relu.cpp), tests/parity_cases.py case_swk_generate_splitmix. */
#ifndef HCONV_TEST_HOOKS_H
#define HCONV_TEST_HOOKS_H
#include "hconv.h"
#ifdef __cplusplus
extern "C" {
#endif
/* TEST HARNESS: the same key structure with the uniform rows of the test oracle's generator (counter-based splitmix64 of seed + 0x1000 + 64 digit + limb) and the
 * per-digit errors e_host[beta][N] (signed 64-bit, HOST) supplied by the caller: lets the product host replay, bit for bit, a network the oracle evaluated under its keys */
int hc_swk_generate_splitmix(hc_ctx *ctx, uint64_t key_id, int level, uint64_t galEl, const uint64_t *sk_ntt, uint64_t seed, const int64_t *e_host);
#ifdef __cplusplus
}
#endif
#endif
