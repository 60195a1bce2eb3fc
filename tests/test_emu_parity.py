"""CPU-side check of the HIP kernel SOURCES: optimal_conv_amd/csrc is compiled with g++ against the fiber emulator
in tests/kernel_emu (threads = ucontext fibers, __syncthreads = yield) and driven through the same C ABI and the
same parity cases as the GPU tests. This validates indexing, LDS exchanges, barrier placement and the host
orchestration without a GPU; it says nothing about gfx950 code generation - tests/test_gpu_a_parity.py does that.
The emulated library is test infrastructure and is never loaded by the package itself."""
import os
import subprocess

import pytest

import parity_cases as pc
from optimal_conv_amd import Context, HconvError
from oracle_lib import Oracle, P0, Q0, Q1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_emu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libhconv_emu.so")


@pytest.fixture(scope="module")
def env():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    ctx = Context([Q0, Q1], [P0], lib_path=EMU_LIB)
    yield ctx, Oracle()
    ctx.close()


def test_ntt(env):
    pc.case_ntt(*env)


def test_pointwise(env):
    pc.case_pointwise(*env)


def test_permute(env):
    pc.case_permute(*env)


def test_const_for(env):
    pc.case_const_for(*env)


def test_rescale(env):
    pc.case_rescale(*env)


def test_keyswitch(env):
    pc.case_keyswitch(*env)


def test_modup_overflow_branch(env):
    pc.case_modup_overflow(*env)


def test_conv_phases(env):
    pc.case_conv_phases(*env)


@pytest.mark.parametrize("max_ob,chunk", [(1, None), (2, None), (4, 1), (8, 3), (16, 32)])
def test_conv_then_pack(env, max_ob, chunk):
    pc.case_conv(*env, max_ob, chunk=chunk)


@pytest.mark.parametrize("max_ob,chunk", [(4, 1), (16, 32)])
def test_conv_then_pack_without_the_small_level_kernels(env, max_ob, chunk):
    """trees of up to 16 nodes run on the 1024-thread S kernels by default (round 3); with small_levels = 0 the same trees go through the
    256-thread kernels of the big levels (b1 .. b4, b5m): both must give the oracle's bits"""
    env[0].set_option("small_levels", 0)
    try:
        pc.case_conv(*env, max_ob, chunk=chunk)
    finally:
        env[0].set_option("small_levels", 16)


@pytest.mark.parametrize("k,i_batch", [(3, 0), (5, 1), (7, 2)])
def test_prep_ker(env, k, i_batch):
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"ref_trace_conv_{k}_{i_batch}.json")
    pc.case_prep_ker(*env, k=k, i_batch=i_batch, trace=json.load(open(path)) if os.path.exists(path) else None)


@pytest.mark.parametrize("max_ob,norm,out_scale", [(8, 2, 2.0 ** 30), (8, 4, 2.0 ** 30), (4, 1, 2.0 ** 43), (8, 8, 2.0 ** 30)])
def test_conv_sparse_norm_and_relu_scale(env, max_ob, norm, out_scale):
    pc.case_conv(*env, max_ob, norm=norm, out_scale=out_scale)


@pytest.mark.parametrize("max_ob,n,chunk,shared", [(4, 3, 64, False), (8, 2, 3, True), (2, 5, 4, False)])
def test_conv_batch(env, max_ob, n, chunk, shared):
    """hc_conv_then_pack_batch (n ciphertexts per launch set) == n separate convolutions == the oracle"""
    pc.case_conv_batch(*env, max_ob, n, chunk=chunk, shared_ker=shared)


@pytest.mark.parametrize("max_ob,G", [(8, 2), (8, 4), (4, 4)])
def test_conv_sharded_over_contexts(env, max_ob, G):
    """hc_conv_then_pack_sharded over G contexts (the C-ABI form of BASELINE config 3's decomposition) on the emulated kernels"""
    pc.case_conv_sharded_abi(lambda: Context([Q0, Q1], [P0], lib_path=EMU_LIB), env[1], max_ob, G)


def test_lv_mul_sum(env):
    pc.case_lv_mul_sum(*env)


def test_encode_slots(env):
    """the slot encoder's kernels (fp64 special FFT without contraction, rounding, NTT) under the emulator vs the oracle"""
    pc.case_encode_slots(*env)


def test_keyswitch_general():
    """general hybrid key switch (BL: level 1, two P primes; bootstrapping shapes) on the emulated kernels"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_keyswitch_general(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), lambda Q, P: Oracle(q=Q, p=P), shapes=((1, 2), (0, 1), (2, 2), (3, 2), (4, 5)))


def test_bl_baseline_conv():
    """scope row 8f-2: the slot-packed baseline's evalConv_BN_BL_test on the C ABI vs the oracle, bit for bit"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_bl_conv(lambda Q, P: Context(Q, P, lib_path=EMU_LIB))


@pytest.mark.parametrize("rows4", [False, True], ids=["rows8", "rows4"])
def test_keyswitch_general_vs_reference_relu_trace(rows4):
    """hc_keyswitch (28 Q + 5 P moduli loaded) vs the reference binary's digests for the bootstrapping chain's key
    switches; the emulator replays the cheap low levels and one five-digit call, the GPU test replays all of them. rows4: the chain's ~30-bit limbs held as 4-byte words"""
    import json
    from oracle_lib import sha_rows
    from test_oracle_pin_keyswitch import ks_inputs
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_trace_ks_relu_5_1.json")))
    Q, P = d["ks_Q"], d["ks_P"]
    ctx = Context(Q, P, lib_path=EMU_LIB)
    if rows4:
        ctx.set_option("pack32", 2)
        assert sum(ctx.row32()) == 11
    for e in [d["events"][1]] + d["events"][-3:]:
        assert e["alpha"] == 5
        cx, evk = ks_inputs(d["seed"], e["call"], e["evk"], e["level"], Q, P, d["N"])
        ctx.swk_load(7, e["level"], evk)
        d0, d1 = ctx.keyswitch(7, e["level"], cx)
        assert sha_rows(*d0) == e["p0"]["sha256"] and sha_rows(*d1) == e["p1"]["sha256"], f"level {e['level']}"
    ctx.close()


def test_ckks_leveled_ops():
    """scope row 8f-1 groundwork: leveled evaluator operations (multi-limb MulRelin + general Rescale, rotation, conjugation,
    MultByi, AddConst, the bootstrapping modulus raise to level 27) composed from C-ABI calls vs the oracle, bit for bit"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_ckks_ops(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), levels=((6, 2.0 ** 30),))


def test_conv_1024_channels_sparse_tile_local_galois(env):
    """the resnet's 8x8 layers: max_ob = 1024 at norm 16; the pack tree then rotates by 2^7+1 and 2^8+1, which permute inside
    the 4096-coefficient tile a b5 workgroup holds in LDS but not inside one 256-coefficient row"""
    pc.case_keyswitch(*env, gals=(129, 257))
    pc.case_conv(*env, 1024, norm=16, out_scale=2.0 ** 41)


def test_keyswitch_qp_mod_down():
    """the halves of the key switch (inner product in QP; ModDownSplitNTTPQ) and arithmetic on QP rows: the reference's BSGS linear transform is made of them"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_keyswitch_qp_mod_down(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), lambda Q, P: Oracle(q=Q, p=P))


def test_keyswitch_hoisted():
    """one digit decomposition shared by several key switches (RotateHoisted), bit-identical to the plain key switch"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_keyswitch_hoisted(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), lambda Q, P: Oracle(q=Q, p=P))


def _rows4(Q, P):
    """a context whose leveled operands hold the ~30-bit limbs' rows as 4-byte words (include/hconv.h, option pack32 = 2: what the bootstrapping host enables)"""
    ctx = Context(Q, P, lib_path=EMU_LIB)
    ctx.set_option("pack32", 2)
    assert any(ctx.row32()) and not all(ctx.row32())
    return ctx


@pytest.mark.parametrize("rows4", [False, True])
def test_leveled_entry_points_row_by_row(rows4):
    """every numpy-in / numpy-out leveled entry point against the oracle's row functions, with 8-byte rows and with 4-byte rows for the small limbs"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_leveled_rows(_rows4 if rows4 else (lambda Q, P: Context(Q, P, lib_path=EMU_LIB)), lambda Q, P: Oracle(q=Q, p=P))


def test_keyswitch_with_seven_and_nine_digits():
    """the bootstrapping chain with TWO special primes: levels 13 and 16 decompose into 7 and 9 digits - more than the 6 after which hc_k_ks_mac_all folds its 128-bit sums"""
    import oracle_ckks
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_keyswitch_general(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), lambda Q, P: Oracle(q=Q, p=P), shapes=((13, 2), (16, 2)), chain=(list(oracle_ckks.Q_SET6), list(oracle_ckks.P_SET6)))


def test_transform_bodies_64_bit_for_the_small_limbs():
    """option small32 = 0: the ~30-bit limbs go through the 64-bit bodies of the batched transforms (the default sends them through the 32-bit bodies, which every other test
    here exercises): the same residues as the oracle either way"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])

    def mk(Q, P):
        ctx = Context(Q, P, lib_path=EMU_LIB)
        ctx.set_option("small32", 0)
        return ctx
    mo = lambda Q, P: Oracle(q=Q, p=P)
    pc.case_keyswitch_general(mk, mo, shapes=((3, 2), (4, 5)))
    pc.case_leveled_rows(mk, mo)


@pytest.mark.parametrize("wgs", [0, 1 << 30], ids=["16-row kernels only", "quarter tiles always"])
def test_batched_transforms_on_quarter_tiles_or_not(wgs):
    """option small_mm_wgs (round 6): a batched inverse pass / second forward pass of few workgroups runs on quarter tiles (hc_k_*_mm_s: four residues per thread). The default
    (1 024 workgroups) puts most launches of the small test shapes on them; here every launch takes ONE of the two forms - the oracle's residues either way, in both row formats."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])

    def mk(Q, P, pack=1):
        ctx = Context(Q, P, lib_path=EMU_LIB)
        ctx.set_option("small_mm_wgs", wgs)
        return ctx
    mo = lambda Q, P: Oracle(q=Q, p=P)
    pc.case_keyswitch_general(mk, mo, shapes=((3, 2), (4, 5)))
    pc.case_leveled_rows(mk, mo)

    def mk2(Q, P):
        ctx = mk(Q, P); ctx.set_option("pack32", 2)
        return ctx
    pc.case_keyswitch_general(mk2, mo, shapes=((4, 3),))


def test_key_switch_with_unpacked_rows():
    """option pack32 = 0 (advertised, untested): keys and digits stay 8-byte words in Montgomery form, the inner products take their generic path (pk = 0)"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])

    def mk(Q, P):
        ctx = Context(Q, P, lib_path=EMU_LIB)
        ctx.set_option("pack32", 0)
        return ctx
    mo = lambda Q, P: Oracle(q=Q, p=P)
    pc.case_keyswitch_general(mk, mo, shapes=((3, 2), (4, 5)))
    pc.case_keyswitch_hoisted(mk, mo)


@pytest.mark.parametrize("case", ["general", "hoisted", "qp"])
def test_key_switch_with_four_byte_rows(case):
    """the key-switch cases above, unchanged, on a context in pack32 = 2: the binding converts at the boundary, the residues are the oracle's"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    mo = lambda Q, P: Oracle(q=Q, p=P)
    if case == "general":
        pc.case_keyswitch_general(_rows4, mo, shapes=((3, 2), (4, 3), (4, 5)))
    elif case == "hoisted":
        pc.case_keyswitch_hoisted(_rows4, mo)
    else:
        pc.case_keyswitch_qp_mod_down(_rows4, mo)


@pytest.mark.parametrize("n,level,alpha", [(3, 4, 3), (2, 5, 2)])
def test_batched_leveled_entry_points(n, level, alpha):
    """hc_set_batch: n images per launch through every leveled entry point == n single-image calls, bit for bit; every fused entry point also == the CPU oracle; a batch
    scope left by an exception leaves the context at one image per call; image strides below an operand's footprint are refused"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_batched_leveled(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), n=n, level=level, alpha=alpha, make_oracle=lambda Q, P: Oracle(q=Q, p=P))


def test_swk_generate_switches_keys():
    """harness key generation on the device: the generated keys satisfy the RLWE key-switching relation (rotation, conjugation, relinearisation)"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_swk_generate(lambda Q, P: Context(Q, P, lib_path=EMU_LIB))


def test_swk_generate_splitmix_equals_the_oracle_generator():
    """test harness: the oracle's key generator reproduced on the device (for replaying the oracle's encrypted network in the product host)"""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    pc.case_swk_generate_splitmix(lambda Q, P: Context(Q, P, lib_path=EMU_LIB), lambda Q, P: Oracle(q=Q, p=P))


def test_free_into_a_foreign_context_is_refused():
    """cached allocations (option async_alloc = 1; the library itself reads no environment variable): a block goes back to the context it came from; handing it to another context's hc_free is an error, not a silent
    hipFree that leaves the owner's block table stale (the lifetime bug behind round 2's synchronising hc_free)"""
    from optimal_conv_amd.abi import DevBuf
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    a, b = Context([Q0, Q1], [P0], lib_path=EMU_LIB), Context([Q0, Q1], [P0], lib_path=EMU_LIB)
    a.set_option("async_alloc", 1); b.set_option("async_alloc", 1)
    buf = DevBuf(a, 1 << 16)
    with pytest.raises(HconvError, match="right after hc_ctx_create"):       # the allocation mode follows the blocks: not while the context owns one
        a.set_option("async_alloc", 0)
    assert b.L.hc_free(b.h, buf.ptr) != 0 and b"not allocated by this context" in b.L.hc_last_error(b.h)
    assert a.L.hc_free(a.h, buf.ptr) == 0
    again = DevBuf(a, 1 << 16)                      # the parked block is handed out again
    assert again.ptr.value == buf.ptr.value
    again.free(); a.close(); b.close()


def test_the_library_reads_no_environment_and_refuses_more_than_five_special_primes(monkeypatch):
    """hc_ctx_create takes nothing from the environment (a cgo host would inherit its shell's); np = 6..8 had a spilling, 30 % slower extension pass
    and is refused (HC_ERR_UNSUPPORTED) instead of shipped; the source holds no getenv at all."""
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, EMU_LIB])
    src = open(os.path.join(ROOT, "optimal_conv_amd", "csrc", "hconv.hip")).read()
    assert "getenv" not in src
    for v in ("HCONV_PACK32", "HCONV_SMALL32", "HCONV_ROT_FUSE", "HCONV_ASYNC_ALLOC"):
        monkeypatch.setenv(v, "0")
    c = Context([0x3FFC0001, 0x40080001, 0x3FAC0001, pc.Q_MIX[4]], pc.P_CHAIN[:5], lib_path=EMU_LIB)
    assert [int(c.L.hc_row_is32(c.h, l)) for l in range(4)] == [0, 0, 0, 0]    # HCONV_PACK32 in the environment changes nothing
    c.set_option("pack32", 2)
    assert [int(c.L.hc_row_is32(c.h, l)) for l in range(4)] == [0, 0, 1, 0]    # limbs 0 and 1 keep 8-byte rows whatever their size (the conv path, Rescale's level-1 branch, sk rows read them so)
    c.close()
    six = pc.P_CHAIN + [0x1FFFFFFFFF380001]
    with pytest.raises(HconvError, match="at most 5"):
        Context(pc.Q_MIX[:3], six, lib_path=EMU_LIB)
