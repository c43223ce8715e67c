"""GPU: CTC prefix beam search (cocr_ctc_beam) against oracle/ctc_ref.py::beam_decoder.  The label sequences and start
frames must be identical; scores are float32 logaddexp chains whose device / numpy libm may differ in the last ulps, so
random cases allow a mismatch only where the oracle's own top-2 final scores are closer than 1e-4 (none observed)."""
import numpy as np
import pytest
import torch

from oracle.ctc_ref import beam_decoder as ref_beam, greedy_decoder as ref_greedy

pytestmark = pytest.mark.gpu


def _engine():
    from conformer_ocr_amd.ctc_decoder import _scratch_engine
    return _scratch_engine(torch.device('cuda', 0))


def _onehot(path, C=6, hi=12.0, lo=-6.0):
    m = np.full((len(path), C), lo, dtype=np.float32)
    for t, c in enumerate(path):
        m[t, c] = hi + 0.1 * t
    return m


def test_beam_equals_greedy_on_peaked_input():
    path = [0, 3, 3, 0, 3, 5, 5, 0, 1, 0]
    m = _onehot(path)
    got = _engine().ctc_beam(torch.from_numpy(m[None]).cuda(), [len(path)], 16)[0]
    want = ref_beam(m.T, 16)
    assert [x[:3] for x in got] == [x[:3] for x in want] == [x[:3] for x in ref_greedy(m.T)]
    np.testing.assert_allclose([x[3] for x in got], [x[3] for x in want], rtol=1e-5)


def test_beam_sums_alignments():
    p = np.array([[0.4, 0.35, 0.25], [0.4, 0.35, 0.25]], dtype=np.float32)
    got = _engine().ctc_beam(torch.from_numpy(np.log(p)[None]).cuda(), [2], 16)[0]
    assert [x[0] for x in got] == [1]          # greedy would output nothing: blank is the per-frame argmax


@pytest.mark.parametrize('C,T,beam', [(5, 12, 4), (11, 40, 16), (32, 60, 16), (128, 48, 16), (93, 33, 8)])
def test_beam_random_matches_oracle(C, T, beam):
    g = np.random.default_rng(C * 1000 + T)
    N = 4
    logits = (g.normal(size=(N, T, C)) * 2.5).astype(np.float32)
    logits[:, :, 0] += 1.5
    lens = [T, T - 3, max(1, T // 2), 1]
    got = _engine().ctc_beam(torch.from_numpy(logits).cuda(), lens, beam)
    for n in range(N):
        want = ref_beam(logits[n, :lens[n]].T, beam)
        assert [x[:2] for x in got[n]] == [x[:2] for x in want], (n, got[n], want)
        assert [x[2] for x in got[n]] == [x[2] for x in want]
        np.testing.assert_allclose([x[3] for x in got[n]], [x[3] for x in want], rtol=1e-4)


def test_beam_zero_length_and_model_attribute(case):
    got = _engine().ctc_beam(torch.zeros((2, 5, 7)).cuda(), [0, 5], 16)
    assert got[0] == []
    # the host class routes a BeamDecoder attribute to the device kernel
    from conformer_ocr_amd.codec import ascii_codec
    from conformer_ocr_amd.ctc_decoder import BeamDecoder
    from conformer_ocr_amd.pred import PytorchRecognitionModel
    hp, state, image, lens, g = case('tiny')
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1,
                                  conv_dropout_p=0.1, codec=ascii_codec(hp.num_classes), ctc_decoder=BeamDecoder(16), compute_dtype='fp32')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0')
    recs = net.predict_labels(torch.from_numpy(image).cuda(), torch.from_numpy(lens))
    for n in range(image.shape[0]):
        want = ref_beam(g['logits'][n, :int(g['out_lens'][n])].T, 16)
        assert [x[0] for x in recs[n]] == [x[0] for x in want]


@pytest.mark.parametrize('C,T,beam,scale', [(128, 300, 16, 2.5), (128, 300, 16, 0.3), (200, 120, 32, 1.5), (17, 90, 16, 1.0), (3, 50, 8, 1.0),
                                              (7, 500, 32, 1.0), (40, 1500, 16, 2.0), (256, 40, 16, 2.0), (2, 30, 4, 1.0)])
def test_beam_pruned_equals_exhaustive(C, T, beam, scale, monkeypatch):
    """cocr_ctc_beam ranks a static pruned candidate set per frame (ctc_beam_rank_kernel + ctc_beam_walk_kernel); COCR_BEAM_REF=1 selects
    the kernel that ranks all beam x C of them.  (7, 500, 32): back-pointers in more than 64 KB of LDS; (40, 1500, 16): in global memory;
    256 and 2 classes: the ends of the fast path's range.  Every output field must agree exactly (same arithmetic, same tie rules), on peaked and on
    nearly flat logits (many near-ties), full-length and ragged lines."""
    from conformer_ocr_amd.engine import HipRecognizer
    from conformer_ocr_amd.spec import HParams
    hp = HParams(num_classes=2, height=16, encoder_dim=16, num_encoder_layers=1, num_attention_heads=1, conv_kernel_size=3,
                 subsampling_conv_channels=8)
    g = np.random.default_rng(C * 7919 + T)
    N = 6
    logits = (g.normal(size=(N, T, C)) * scale).astype(np.float32)
    logits[:, :, 0] += 1.0
    logits[1, :, 1:] = np.round(logits[1, :, 1:] * 2) / 2          # exact ties between classes
    lens = [T, T - 1, T // 2, 1, T, 7]
    x = torch.from_numpy(logits).cuda()
    fast = HipRecognizer(hp, torch.device('cuda', 0), 'fp32').ctc_beam(x, lens, beam)
    monkeypatch.setenv('COCR_BEAM_REF', '1')
    ref = HipRecognizer(hp, torch.device('cuda', 0), 'fp32').ctc_beam(x, lens, beam)
    assert fast == ref


def test_metric_model_forward_then_beam16_against_the_oracle(text_case):
    """BASELINE configs[4] composed: the metric's model (cfg2_text: D=256, 12 blocks) on 96x1200 lines -> HIP logits (T = 300,
    128 classes) -> cocr_ctc_beam(beam 16), against oracle/ctc_ref.py::beam_decoder run on the SAME logits: labels, start and end
    frames exact, scores within 1e-4 relative.  fp32 logits of lines 0-3 (the latency-bound small batch) and bf16 logits of all 32
    lines; and on these peaked lines the beam's label string is the reference's greedy string.  (Semantics of the beam search are
    the build's own restatement of kraken's algorithm: kraken is absent, the reference never calls it -- parity unpinned.)"""
    from tests.hip_util import make_engine
    tc = text_case('cfg2_text')
    image, lens, idx = tc.batch(0)
    for dtype, n in (('fp32', 4), ('bf16', tc.n)):
        eng = make_engine(tc.hp, tc.state, dtype)
        x = torch.from_numpy(image[:n, 0]).cuda()
        logits, out_lens = eng.forward(x, lens[:n])
        assert logits.shape == (n, 300, 128)
        got = eng.ctc_beam(logits, out_lens, 16)
        host = logits.cpu().numpy()
        for k in range(n):
            want = ref_beam(host[k, :int(out_lens[k])].T, 16)
            assert [r[:3] for r in got[k]] == [r[:3] for r in want], (dtype, k)
            np.testing.assert_allclose([r[3] for r in got[k]], [r[3] for r in want], rtol=1e-4)
            assert [r[0] for r in got[k]] == tc.ref_strings[idx[k]], (dtype, k)
