"""GPU: line pre-processing (cocr_preproc_lines) against oracle/preproc_ref.py -- integer work, bit-exact -- on ragged batches of
grayscale and RGB crops, up- and down-scaling, degenerate sizes, bucketed widths; and end to end into the u8 forward."""
import numpy as np
import pytest
import torch

from oracle import preproc_ref as P

pytestmark = pytest.mark.gpu


def _engine(case):
    from tests.hip_util import make_engine
    hp, state, image, lens, g = case('tiny')
    return make_engine(hp, state, 'fp32'), hp


def _lines(seed, shapes):
    g = np.random.default_rng(seed)
    out = []
    for s in shapes:
        x = g.integers(0, 256, size=s, dtype=np.uint8)
        x[: s[0] // 3] = 255                      # saturated background: Lanczos overshoot must clip like Pillow
        out.append(x)
    return out


@pytest.mark.parametrize('height,pad', [(96, 16), (16, 0), (48, 5)])
def test_preproc_bit_exact(case, height, pad):
    eng, hp = _engine(case)
    lines = _lines(height, [(57, 311), (120, 1500), (height, 400), (200, 777, 3), (31, 64), (1, 9), (300, 5), (143, 999, 3)])
    got, lens = eng.preprocess(lines, height=height, pad=pad, bucket_edge=200)
    want, wlens = P.collate(lines, height, pad, width=got.shape[2])
    assert got.shape[2] % 200 == 0 and lens.tolist() == wlens.tolist()
    assert np.array_equal(got.cpu().numpy(), want)


def test_preproc_errors_and_forward(case):
    eng, hp = _engine(case)
    lines = _lines(3, [(40, 300), (16, 64), (23, 200)])
    with pytest.raises(ValueError, match='size mismatch'):
        eng.preprocess(lines, width=64)
    batch, lens = eng.preprocess(lines)                       # model height (16), pad 16
    want, _ = P.collate(lines, hp.height, 16)
    assert np.array_equal(batch.cpu().numpy(), want)
    logits, out_lens = eng.forward(batch, lens)              # u8 ingest: pixel / 255
    ref_logits, _ = eng.forward(torch.from_numpy(want.astype(np.float32) / 255.0).cuda(), lens)
    assert torch.equal(logits, ref_logits)


def test_host_class_transform_lines(case):
    """PytorchRecognitionModel.transform_lines -> predict_string: raw crops in, strings out."""
    from conformer_ocr_amd.codec import ascii_codec
    from conformer_ocr_amd.pred import PytorchRecognitionModel
    hp, state, image, lens, g = case('tiny')
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1,
                                  conv_dropout_p=0.1, codec=ascii_codec(hp.num_classes), compute_dtype='fp32')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0').eval()
    crops = _lines(11, [(33, 260), (16, 90), (50, 410, 3)])
    batch, l = net.transform_lines(crops, bucket_edge=64)
    assert batch.shape[:3] == (3, 1, hp.height) and batch.shape[3] % 64 == 0 and batch.dtype == torch.uint8
    want, wl = P.collate(crops, hp.height, 16, width=batch.shape[3])
    assert np.array_equal(batch[:, 0].cpu().numpy(), want) and l.tolist() == wl.tolist()
    strings = net.predict_string(batch, l)
    assert len(strings) == 3 and all(isinstance(x, str) for x in strings)
