"""The whole convReLU tail against the reference binary, end to end: `gotrace -chain` planted the input of BootstrappConv_CtoS and every
switching key (by the kind of key switch that reads it) in a `convReLU 5 1 1` run of /root/reference/test_run and recorded the SHA-256 of the
ciphertexts BootstrappConv_CtoS returns (two, level 14), of SlotsToCoeffs' result and of the ciphertext the layer hands on after the last Rescale
(tests/golden/ref_trace_chain_5_1.json). tests/oracle_ckks.py's chain -- modUp, four BSGS linear transforms in the extended basis, conjugation,
the sine (EvaluateCheby + double angles), MultByConst, three EvaluatePoly + the product of evalReLU, MulByPow2, the keep mask, three more linear
transforms, two rescales: 207 key switches, 28 levels -- replayed on the oracle must arrive at the SAME residues."""
import pytest

import chain_replay


@pytest.mark.slow
def test_convrelu_tail_end_to_end_equals_the_reference_binary():
    n, _ = chain_replay.replay()
    assert n == 4


@pytest.mark.slow
def test_sparse_ctos_equals_the_reference_binary():
    """round 3: BootstrappConv_CtoS of the sparse-slot bootstrapper (log_sparse 2; the resnet's btp3, main.go:480-500) against the binary's digests
    on planted data (`gotrace -chain -logslots 13`): subSum, the sparse DFT matrices with their repacking, the packed (re | im) ciphertext through
    the sine - ten checkpoints from modUp to the ciphertext the function ends on"""
    assert chain_replay.replay_sparse() == 10


@pytest.mark.slow
def test_baseline_bootstrapp_equals_the_reference_binary():
    """round 3: the baseline half of convReLU - the stock Bootstrapp on parameter set [7] - against the binary's digests on planted data (`gotrace -flow-bl -chain`):
    SetScale, modUp, seven LinearTransforms, conjugation, CoeffsToSlots, both EvaluateCheby, evaluateSine, SlotsToCoeffs, the returned level-14 ciphertext"""
    assert chain_replay.replay_bl() == 18
