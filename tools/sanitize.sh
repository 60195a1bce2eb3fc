#!/bin/bash
# CPU-side sanitizer runs (GPU ASan is not available on this pool): UBSan and ASan over the kernel SOURCES running under the fiber
# emulator + host orchestration, ASan+UBSan over the oracle. All must print "... RUN OK" with no sanitizer report.
set -eu
cd "$(dirname "$0")/.."
# the instrumentation flags live ONLY here (this script is CPU-only and listed in .gpurunignore); the Makefiles take SAN_FLAGS
FS="-fsanitize"
make -s -C tests/kernel_emu "$PWD/tests/kernel_emu/_build/libhconv_emu_ubsan.so" SAN_FLAGS="$FS=undefined -fno-sanitize-recover=undefined"
make -s -C tests/kernel_emu "$PWD/tests/kernel_emu/_build/libhconv_emu_asan.so" SAN_FLAGS="$FS=address -fno-omit-frame-pointer"
make -s -C oracle asan SAN_FLAGS="$FS=address,undefined -fno-omit-frame-pointer"
cat > /tmp/hc_ubsan_run.py <<PY
import sys
sys.path.insert(0, "$PWD"); sys.path.insert(0, "$PWD/tests")
import parity_cases as pc
from oracle_lib import Oracle, Q0, Q1, P0
from optimal_conv_amd import Context
ctx = Context([Q0, Q1], [P0], lib_path="$PWD/tests/kernel_emu/_build/libhconv_emu_ubsan.so"); O = Oracle()
pc.case_ntt(ctx, O); pc.case_pointwise(ctx, O); pc.case_rescale(ctx, O); pc.case_keyswitch(ctx, O); pc.case_modup_overflow(ctx, O)
pc.case_conv(ctx, O, 8, chunk=3); pc.case_prep_ker(ctx, O, 3, 0)
LIB = "$PWD/tests/kernel_emu/_build/libhconv_emu_ubsan.so"
pc.case_keyswitch_general(lambda Q, P: Context(Q, P, lib_path=LIB), lambda Q, P: Oracle(q=Q, p=P), shapes=((1, 2), (4, 3)))   # batched multi-modulus key switch
pc.case_ckks_ops(lambda Q, P: Context(Q, P, lib_path=LIB), levels=((6, 2.0 ** 30),))                                        # hc_lv_*, general rescale, mod raise
print("UBSAN KERNEL-SOURCE RUN OK")
PY
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libubsan.so) python /tmp/hc_ubsan_run.py
cat > /tmp/hc_asan_run.py <<PY
import sys
sys.path.insert(0, "$PWD/tests")
import oracle_lib
oracle_lib.build = lambda: "$PWD/oracle/liboracle_asan.so"
import numpy as np
from oracle_lib import Oracle, Q0, splitmix_rows
import parity_cases as pc
O = Oracle()
ct_in, ker = pc.planted_conv_inputs(5, 4)
evk = np.zeros((16, 4, 65536), dtype=np.uint64)
for j in (15, 16): evk[j - 1] = pc.seeded_evk(100 + j)
O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk, 4, 1, 2.0 ** 30, splitmix_rows(9, Q0, 65536))
import oracle_ckks as ck
C = ck.Ckks(logN=10, h=64)                                            # general key switch (5 special primes), rescale, relin key generation
ck.conv_relu_tail(C, ck.Bootstrapper(C), C.encrypt_coeffs(np.linspace(-9, 9, C.N), 0, 2.0 ** 43, seed=2), 0.0, 4, 16, 15)
sk = O.gen_sk(1); O.gen_galois_key_l0(sk, 65537, 3); O.encrypt(sk, O.encode_coeffs(np.linspace(-1, 1, 65536), 2.0 ** 30, [0, 1]), 1, 4)
print("ASAN ORACLE RUN OK")
PY
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python /tmp/hc_asan_run.py

# AddressSanitizer over the kernel sources + orchestration under the emulator (device memory = heap blocks), in both allocation modes
cat > /tmp/hc_asan_emu.py <<PY
import sys, os
sys.path.insert(0, "$PWD"); sys.path.insert(0, "$PWD/tests")
import parity_cases as pc
from oracle_lib import Oracle, Q0, Q1, P0
from optimal_conv_amd import Context
LIB = "$PWD/tests/kernel_emu/_build/libhconv_emu_asan.so"
for mode in ("0", "1"):
    os.environ["HCONV_ASYNC_ALLOC"] = mode
    ctx = Context([Q0, Q1], [P0], lib_path=LIB); O = Oracle()
    pc.case_ntt(ctx, O); pc.case_rescale(ctx, O); pc.case_keyswitch(ctx, O)
    pc.case_conv(ctx, O, 8, chunk=3); pc.case_conv(ctx, O, 4); pc.case_prep_ker(ctx, O, 3, 0)
    ctx.close()
    pc.case_keyswitch_general(lambda Q, P: Context(Q, P, lib_path=LIB), lambda Q, P: Oracle(q=Q, p=P), shapes=((1, 2), (4, 3)))
    pc.case_ckks_ops(lambda Q, P: Context(Q, P, lib_path=LIB), levels=((6, 2.0 ** 30),))
    print("ASAN KERNEL-SOURCE RUN OK (HCONV_ASYNC_ALLOC=%s)" % mode)
PY
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python /tmp/hc_asan_emu.py
# instrumented builds never travel to the GPU box
rm -f oracle/liboracle_asan.so tests/kernel_emu/_build/libhconv_emu_asan.so tests/kernel_emu/_build/libhconv_emu_ubsan.so
