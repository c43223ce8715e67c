"""GPU: the evaluation loop (bucketing, batching, predict_string, CER) -- BASELINE configs[3] flavour: mixed widths with
length bucketing on the wide model is too slow for the CPU oracle, so the check here is the metric's own definition:
bf16 strings against fp32 strings of the same HIP path (CER bounded), and fp32 strings against the CPU oracle's greedy
strings on a small sample (identical outside the margin filter is covered in test_hip_parity)."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.codec import ascii_codec
from conformer_ocr_amd.evaluate import evaluate, recognize
from conformer_ocr_amd.pred import PytorchRecognitionModel

pytestmark = pytest.mark.gpu


def _net(hp, state, dtype):
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1,
                                  conv_dropout_p=0.1, codec=ascii_codec(hp.num_classes), compute_dtype=dtype)
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return net.to('cuda:0').eval()


def test_bucketed_evaluation_mixed_widths(case):
    hp, state, *_ = case('cfg1')
    g = np.random.default_rng(5)
    widths = [int(w) for w in g.integers(25, 100, size=24) * 8]           # 200 .. 792 px, step 8
    lines = [synth.make_lines(1, hp.height, w, seed=100 + i)[0][0, 0] for i, w in enumerate(widths)]
    ref = recognize(_net(hp, state, 'fp32'), lines, batch_size=8)
    assert sorted(ref) == list(range(24)) and all(isinstance(s, str) for s in ref.values())
    truths = [ref[i] for i in range(24)]
    rep = evaluate(_net(hp, state, 'bf16'), lines, truths, batch_size=8)
    assert rep['lines'] == 24 and rep['chars'] > 100 and 0.0 <= rep['cer'] <= 1.0
    # (no bound on this CER: random-weight logits have top-2 margins of a few bf16 roundings, so bf16-vs-fp32 strings of THIS model say
    # nothing; the CER of the bf16 path is asserted == 0 against a ground truth on the text fixtures, tests/test_hip_bf16_path.py)
    # rank-count invariance: a 2-rank split yields the same strings for every line (fixed bucket edges)
    a = recognize(_net(hp, state, 'fp32'), lines, batch_size=8, rank=0, world=2)
    b = recognize(_net(hp, state, 'fp32'), lines, batch_size=8, rank=1, world=2)
    assert set(a) | set(b) == set(range(24)) and not (set(a) & set(b))
    assert all(ref[i] == s for i, s in {**a, **b}.items())


def test_pipelined_loop_equals_serial_loop(case):
    """recognize(pipelined=True) overlaps upload / forward / read-back of consecutive batches; the strings must be those of the
    serial loop (also with a beam decoder and with a user-supplied host decoder)."""
    from conformer_ocr_amd.ctc_decoder import BeamDecoder
    hp, state, *_ = case('cfg1')
    g = np.random.default_rng(6)
    widths = [int(w) for w in g.integers(25, 90, size=20) * 8]
    lines = [synth.make_lines(1, hp.height, w, seed=300 + i)[0][0, 0] for i, w in enumerate(widths)]
    net = _net(hp, state, 'fp32')
    serial = recognize(net, lines, batch_size=4, pipelined=False)
    assert recognize(net, lines, batch_size=4, pipelined=True) == serial
    assert recognize(net, lines, batch_size=4, streams=3) == serial          # three batches in flight on three streams / model copies
    net.ctc_decoder = BeamDecoder(4)
    beam_serial = recognize(net, lines, batch_size=4, pipelined=False)
    assert recognize(net, lines, batch_size=4, pipelined=True) == beam_serial
    assert recognize(net, lines, batch_size=4, streams=4) == beam_serial
    from oracle.ctc_ref import greedy_decoder as host_greedy
    net.ctc_decoder = host_greedy                     # any callable (ncls, T) -> records: the reference's host loop
    assert recognize(net, lines, batch_size=4, pipelined=True) == serial


def test_recognize_from_raw_crops(case):
    """recognize_crops = GPU pre-processing + the recognition loop: same strings as `recognize` on the lines the pre-processing
    oracle produces from the same crops."""
    from conformer_ocr_amd.evaluate import recognize_crops
    from oracle import preproc_ref as P
    hp, state, *_ = case('cfg1')
    g = np.random.default_rng(17)
    crops = []
    for i in range(14):
        h, w = int(g.integers(40, 160)), int(g.integers(200, 900))
        c = g.integers(0, 256, size=(h, w) if i % 3 else (h, w, 3), dtype=np.uint8)
        crops.append(c)
    net = _net(hp, state, 'fp32')
    got = recognize_crops(net, crops, batch_size=4)
    lines = [P.preprocess_line(c, hp.height, 16).astype(np.float32) / 255.0 for c in crops]
    assert got == recognize(net, lines, batch_size=4)


def test_validation_step_loss_and_metrics(case):
    """`net.step` = the reference's `_step` (forward + CTC criterion): the loss equals the CPU restatement's on the probits the
    forward returned; `validate` reports the reference's validation-epoch numbers (CER / WER of greedy strings, mean batch loss)."""
    from conformer_ocr_amd.evaluate import validate
    from oracle import ctc_loss_ref as R
    hp, state, image, lens, _ = case('cfg1')
    net = _net(hp, state, 'fp32')
    g = np.random.default_rng(3)
    N = image.shape[0]
    label_lens = g.integers(0, 12, size=N)
    target = np.concatenate([g.integers(1, hp.num_classes, size=l) for l in label_lens] + [np.zeros(0, np.int64)])
    o = net.step({'image': torch.from_numpy(image).cuda(), 'seq_lens': torch.from_numpy(lens), 'target': torch.from_numpy(target),
                  'target_lens': torch.from_numpy(label_lens)}, with_grad=True)
    probits = o['probits'].cpu().numpy()
    want_nll, want_grad = R.ctc_loss(probits, target, o['output_lens'].numpy(), label_lens)
    np.testing.assert_allclose(o['nll'].cpu().numpy(), want_nll, rtol=2e-6, atol=1e-3)
    assert abs(float(o['loss']) - want_nll.sum()) < 2e-6 * want_nll.sum() + 1e-3
    np.testing.assert_allclose(o['grad_probits'].cpu().numpy(), want_grad, atol=2e-3)
    # validation epoch on mixed widths: the truths are the model's own strings, except two lines edited by hand
    widths = [int(w) for w in g.integers(25, 80, size=12) * 8]
    lines = [synth.make_lines(1, hp.height, w, seed=700 + i)[0][0, 0] for i, w in enumerate(widths)]
    own = recognize(net, lines, batch_size=4)
    truths = [own[i] for i in range(12)]
    rep = validate(net, lines, truths, batch_size=4)
    assert rep['cer'] == 0.0 and rep['val_accuracy'] == 1.0 and rep['batches'] >= 3 and rep['val_loss'] > 0
    truths[0] = truths[0][1:]
    truths[5] = truths[5] + 'xy'
    rep2 = validate(net, lines, truths, batch_size=4)
    assert abs(rep2['cer'] - 3 / rep2['chars']) < 1e-12 and rep2['val_loss'] != rep['val_loss']


def test_pinned_staging_pool_shares_buffers_between_batch_shapes():
    """The loop's pinned staging memory: one size class serves every batch shape that fits it, a buffer comes back with the event of
    the copy that read it, and a consumer that never gives its buffers back does not stall the next taker for ever."""
    from conformer_ocr_amd.evaluate import _PinnedPool
    pool = _PinnedPool()
    flat_a, a = pool.take((3, 1, 16, 100), torch.float32, depth=2)
    assert a.shape == (3, 1, 16, 100) and a.dtype == torch.float32 and a.is_pinned() and flat_a.numel() == 1 << 20
    a.fill_(0.5)
    dev = a.to('cuda:0', non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    pool.give(flat_a, ev)
    flat_b, b = pool.take((2, 1, 16, 50), torch.uint8, depth=1)            # another shape and type, the same 1 MiB class: the same memory
    assert flat_b.data_ptr() == flat_a.data_ptr() and b.shape == (2, 1, 16, 50) and b.dtype == torch.uint8
    assert float(dev.min()) == 0.5                                           # (the copy had finished before the buffer was handed out again)
    flat_c, _ = pool.take((2, 1, 16, 50), torch.uint8, depth=1)            # flat_b was never given back: a fresh buffer after the wait
    assert flat_c.data_ptr() != flat_b.data_ptr()
    assert _PinnedPool.size_class(3 << 20) == 4 << 20 and _PinnedPool.size_class(1) == 1 << 20


def test_configs2_sharding_rehearsed_on_one_gpu():
    """BASELINE configs[2]: batch = 256 lines of 96x1200 sharded over 8 ranks as independent 32-line batches.  Rehearsed on the one GPU
    of the box: the eight ranks' shards (`recognize(..., rank=r, world=8)`, the same code path a rank runs; `dist.shard_batches`) are
    disjoint, cover all 256 lines, give 32 lines each, and every line's string equals the single-rank run's (a line's logits depend on
    its padded batch only, the batches are the same in every world size) -- the metric's model itself (12 blocks)."""
    hp = synth.hparams('cfg2')
    state = synth.make_state_dict(hp, seed=21, decoder_gain=8.0, style='text')
    net = _net(hp, state, 'bf16')
    lines = [synth.make_text_lines(1, hp.height, 1200, seed=4000 + i)[0][0, 0] for i in range(256)]
    whole = recognize(net, lines, batch_size=32, streams=4)
    assert sorted(whole) == list(range(256))
    seen = {}
    for r in range(8):
        part = recognize(net, lines, batch_size=32, rank=r, world=8, streams=2)
        assert len(part) == 32 and not (set(part) & set(seen))
        seen.update(part)
    assert seen == whole
    assert len(set(whole.values())) > 100             # (the lines read differently: the comparison is not between empty strings)
