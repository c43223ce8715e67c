"""GPU: the CTC greedy kernel through the C ABI against the CPU restatement -- integer fields exact."""
import numpy as np
import pytest
import torch

from oracle.ctc_ref import greedy_decoder as ref_greedy

pytestmark = pytest.mark.gpu


def _engine():
    from conformer_ocr_amd.ctc_decoder import _scratch_engine
    return _scratch_engine(torch.device('cuda', 0))


def _compare(logits, lens):
    eng = _engine()
    got = eng.ctc_greedy(torch.from_numpy(logits).cuda(), lens)
    for n in range(logits.shape[0]):
        want = ref_greedy(logits[n, :lens[n]].T)
        assert [x[:3] for x in got[n]] == [x[:3] for x in want], n
        np.testing.assert_array_equal(np.float32([x[3] for x in got[n]]), np.float32([x[3] for x in want]))


def test_greedy_known_answer():
    path = [0, 3, 3, 0, 3, 5, 5, 0]
    m = np.full((1, len(path), 6), -1.0, np.float32)
    for t, c in enumerate(path):
        m[0, t, c] = 5.0 + 0.1 * t
    got = _engine().ctc_greedy(torch.from_numpy(m).cuda(), [len(path)])[0]
    assert [x[:3] for x in got] == [(3, 1, 2), (3, 4, 4), (5, 5, 6)]


@pytest.mark.parametrize('ncls', [2, 11, 64, 93, 128, 300])
def test_greedy_random_with_ties_and_ragged_lengths(ncls):
    g = np.random.default_rng(ncls)
    N, T = 7, 301
    logits = g.integers(-3, 4, (N, T, ncls)).astype(np.float32)      # few distinct values: many exact ties
    logits[:, :, 0] += (g.uniform(size=(N, T)) < 0.3) * 4.0
    lens = [T, 0, 1, 2, 64, 65, 300]
    _compare(logits, lens)


def test_greedy_on_golden_logits(case):
    hp, state, image, lens, g = case('cfg2')
    logits = g['logits_head']
    _compare(np.ascontiguousarray(logits), [300, 300, 17, 299])


def test_greedy_all_blank_and_single_run():
    m = np.zeros((2, 50, 9), np.float32)
    m[0, :, 0] = 1.0
    m[1, :, 4] = 1.0
    got = _engine().ctc_greedy(torch.from_numpy(m).cuda(), [50, 50])
    assert got[0] == [] and [x[:3] for x in got[1]] == [(4, 0, 49)]


def test_callable_decoder_object():
    from conformer_ocr_amd.ctc_decoder import greedy_decoder
    m = np.random.default_rng(0).normal(size=(11, 40)).astype(np.float32)
    assert [x[:3] for x in greedy_decoder(m)] == [x[:3] for x in ref_greedy(m)]


def test_argmax_in_the_decoder_epilogue_equals_the_argmax_kernel():
    """Up to 128 classes the decoder product's epilogue leaves the per-frame argmax / maximum of the logits it writes; cocr_ctc_greedy on
    those logits only merges runs.  Same records as the stand-alone argmax kernel (cocr_forget_argmax), including exact ties (first
    index) and after an in-place edit of the logits (the Python wrapper notices the tensor's version and decodes from the values)."""
    import torch
    from conformer_ocr_amd import synth
    from tests.hip_util import make_engine
    hp = synth.hparams('cfg2', num_encoder_layers=1)
    state = synth.make_state_dict(hp, seed=21, decoder_gain=8.0)
    image, lens = synth.make_lines(5, hp.height, 333, seed=4, widths=[333, 100, 250, 17, 300])
    x = torch.from_numpy(image[:, 0]).cuda()
    for dtype in ('bf16', 'fp32'):
        for ties in (False, True):
            st = dict(state)
            if ties:                                             # every frame: classes 5 and 9 tie for the maximum, then 0 and 3
                st['decoder.weight'] = np.zeros_like(state['decoder.weight'])
                b = np.full(hp.num_classes, -1.0, np.float32)
                b[[5, 9]] = 2.0
                st['decoder.bias'] = b
            eng = make_engine(hp, st, dtype)
            lg, ol = eng.forward(x, lens)
            fused = eng.ctc_greedy(lg, ol)
            eng.lib.cocr_forget_argmax(eng._h)
            plain = eng.ctc_greedy(lg, ol)
            assert fused == plain
            if ties:
                assert all(r == [(5, 0, int(ol[n]) - 1, 2.0)] for n, r in enumerate(fused) if ol[n] > 0)
            lg2, ol = eng.forward(x, lens)
            lg2[:, :, 7] += 100.0                                # in place: class 7 now wins every frame
            edited = eng.ctc_greedy(lg2, ol)
            assert all(r[0][0] == 7 and len(r) == 1 for n, r in enumerate(edited) if ol[n] > 0)
