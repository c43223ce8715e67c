"""The line pre-processing oracle (oracle/preproc_ref.py) pinned against Pillow itself: its 8-bit LANCZOS resize and RGB->L
conversion must agree bit for bit with PIL on up-scaling, down-scaling, identity and degenerate sizes."""
import numpy as np
import pytest

from oracle import preproc_ref as P

PIL = pytest.importorskip('PIL.Image')


@pytest.mark.parametrize('h,w,oh', [(57, 311, 96), (120, 1500, 96), (96, 400, 96), (200, 777, 96), (31, 64, 96), (143, 999, 48),
                                    (1, 9, 96), (300, 5, 96), (97, 1203, 96)])
def test_resize_matches_pillow(h, w, oh):
    g = np.random.default_rng(h * 10007 + w)
    img = g.integers(0, 256, size=(h, w), dtype=np.uint8)
    if (h + w) % 2:
        img[:, : w // 2] = 255           # saturated regions: the Lanczos lobes overshoot and must clip like Pillow's
    ow = P.scaled_width(h, w, oh)
    ref = np.asarray(PIL.fromarray(img, 'L').resize((ow, oh), PIL.Resampling.LANCZOS))
    assert np.array_equal(P.resize_lanczos_u8(img, ow, oh), ref)


def test_rgb_to_l_matches_pillow():
    g = np.random.default_rng(5)
    img = g.integers(0, 256, size=(40, 70, 3), dtype=np.uint8)
    assert np.array_equal(P.rgb_to_l(img), np.asarray(PIL.fromarray(img, 'RGB').convert('L')))


def test_preprocess_and_collate_contract():
    g = np.random.default_rng(9)
    lines = [g.integers(0, 256, size=(h, w), dtype=np.uint8) for h, w in [(60, 500), (96, 300), (130, 900)]]
    batch, lens = P.collate(lines, 96, 16)
    assert lens.tolist() == [int(500 * 96 / 60) + 32, 300 + 32, int(900 * 96 / 130) + 32]
    assert batch.shape == (3, 96, int(lens.max())) and batch.dtype == np.uint8
    assert not batch[1, :, :16].any() and not batch[1, :, lens[1] - 16:].any()          # 16 px of padding left and right, zeros to the batch width
    assert np.array_equal(batch[1, :, 16:316], lines[1])                                 # height 96 already: the resize is the identity
    with pytest.raises(ValueError):
        P.collate(lines, 96, 16, width=100)
