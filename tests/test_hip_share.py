"""cocr_share_weights: packed copies of one model that read one set of weights (include/cocr.h)."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.engine import HipRecognizer

pytestmark = pytest.mark.gpu


def _engines(dtype='bf16'):
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=41, decoder_gain=4.0)
    dev = torch.device('cuda', 0)
    owner = HipRecognizer(hp, dev, dtype)
    owner.load_state(state)
    owner.finalize()
    private = HipRecognizer(hp, dev, dtype)
    private.load_state(state)
    private.finalize()
    sharer = HipRecognizer(hp, dev, dtype)
    sharer.share_weights(owner)
    return hp, state, owner, private, sharer


def test_a_model_that_shares_weights_computes_what_its_owner_computes():
    hp, state, owner, private, sharer = _engines()
    img, lens = synth.make_lines(5, hp.height, 700, seed=3, widths=[700, 650, 300, 40, 512])
    x = torch.from_numpy(img[:, 0]).cuda()
    for eng in (owner, private, sharer):
        eng.set_graph(True)
    outs = []
    for rep in range(3):                                 # plain run, captured run, replay
        outs.append([eng.forward(x, lens)[0].cpu().numpy() for eng in (owner, private, sharer)])
    torch.cuda.synchronize()
    for a, b, c in outs:
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # two sharers on two streams at once
    other = HipRecognizer(hp, owner.device, 'bf16')
    other.share_weights(owner)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        l1, _ = sharer.forward(x, lens)
    with torch.cuda.stream(s2):
        l2, _ = other.forward(x, lens)
    torch.cuda.synchronize()
    assert np.array_equal(l1.cpu().numpy(), outs[0][0]) and np.array_equal(l2.cpu().numpy(), outs[0][0])


def test_a_line_longer_than_the_tables_seen_first_by_the_sharer():
    """The positional tables grow for a long line (embedding.py:35-41) -- here inside the model that shares them; the owner, with launch
    sequences captured on the old tables, must follow."""
    hp, state, owner, private, sharer = _engines()
    for eng in (owner, private, sharer):
        eng.set_graph(True)
    img, lens = synth.make_lines(2, hp.height, 600, seed=5)
    x = torch.from_numpy(img[:, 0]).cuda()
    before = [owner.forward(x, lens)[0].cpu().numpy() for _ in range(3)][-1]          # owner: captured on the first tables
    limg, llens = synth.make_lines(1, hp.height, 4 * 5200, seed=6)                     # 5200 frames > the 4999 the first tables cover
    lx = torch.from_numpy(limg[:, 0]).cuda()
    long_shared = sharer.forward(lx, llens)[0].cpu().numpy()
    long_private = private.forward(lx, llens)[0].cpu().numpy()
    assert np.array_equal(long_shared, long_private)
    after = owner.forward(x, lens)[0].cpu().numpy()
    assert np.array_equal(after, before)
    assert np.array_equal(owner.forward(lx, llens)[0].cpu().numpy(), long_private)


def test_weights_change_through_the_owner_only():
    hp, state, owner, private, sharer = _engines('fp32')
    with pytest.raises(RuntimeError, match='owner'):
        sharer.weight_blob()
    with pytest.raises((RuntimeError, ValueError)):
        sharer.share_weights(sharer)
    with pytest.raises((RuntimeError, ValueError)):
        owner.share_weights(sharer)                      # the owner of a sharer cannot share (sharer is not an owner: it shares itself)
    img, lens = synth.make_lines(2, hp.height, 300, seed=9)
    x = torch.from_numpy(img[:, 0]).cuda()
    a = sharer.forward(x, lens)[0].cpu().numpy()
    state2 = {k: (v * 1.01 if k == 'decoder.weight' else v) for k, v in state.items()}
    owner.load_state(state2)
    owner.finalize()                                     # the owner's buffers move
    private.load_state(state2)
    private.finalize()
    b = sharer.forward(x, lens)[0].cpu().numpy()
    assert not np.array_equal(a, b) and np.array_equal(b, private.forward(x, lens)[0].cpu().numpy())
    sharer.load_state(state)
    sharer.finalize()                                    # weights of its own again
    assert np.array_equal(sharer.forward(x, lens)[0].cpu().numpy(), a)
    del owner                                            # (the sharer no longer depends on it)
    assert np.array_equal(sharer.forward(x, lens)[0].cpu().numpy(), a)


def test_owner_finalized_again_in_another_compute_dtype():
    """The header allows weight changes through the owner (set_tensor + finalize): also a finalize in the OTHER compute dtype.  The
    sharer then follows -- layout, plan, workspace and captured launches re-derived at its next forward -- instead of running its bf16
    kernels over an fp32 blob.  And a sharer hands out no blob copy (its view of the owner's buffer may be gone): the owner does."""
    hp, state, owner, private, sharer = _engines('bf16')
    img, lens = synth.make_lines(3, hp.height, 500, seed=12, widths=[500, 420, 77])
    x = torch.from_numpy(img[:, 0]).cuda()
    sharer.set_graph(True)
    bf = [sharer.forward(x, lens)[0].cpu().numpy() for _ in range(3)][-1]
    assert np.array_equal(bf, private.forward(x, lens)[0].cpu().numpy())
    with pytest.raises(RuntimeError, match='owner'):
        sharer.export_blob()
    owner.compute_dtype = 'fp32'
    owner.finalize()
    p32 = HipRecognizer(hp, owner.device, 'fp32')
    p32.load_state(state)
    p32.finalize()
    want = p32.forward(x, lens)[0].cpu().numpy()
    for _ in range(3):                                   # plain, captured, replayed: all on the new blob
        got = sharer.forward(x, lens)[0].cpu().numpy()
        assert np.array_equal(got, want)
    assert np.abs(got - bf).max() > 1e-4                 # (it is the fp32 result, not the old one)
    owner.compute_dtype = 'bf16'
    owner.finalize()
    assert np.array_equal(sharer.forward(x, lens)[0].cpu().numpy(), bf)
