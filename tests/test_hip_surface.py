"""GPU: the drop-in class notices every way its weights can change, and the fused greedy argmax is bound to the tensor it was
computed for (pred.py `_signature`, engine.py fused-argmax shortcut)."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.codec import ascii_codec
from conformer_ocr_amd.pred import PytorchRecognitionModel
from tests.hip_util import make_engine

pytestmark = pytest.mark.gpu
DROPS = dict(input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1)


def _net(hp, state, dtype='fp32'):
    net = PytorchRecognitionModel(**hp.as_dict(), **DROPS, codec=ascii_codec(hp.num_classes), compute_dtype=dtype)
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return net.to('cuda:0').eval()


def test_every_kind_of_weight_change_reaches_the_packed_model():
    """In-place writes, load_state_dict, load_state_dict(assign=True), `p.data = new`, a replaced parameter, a replaced sub-module
    (`net.nn.decoder = nn.Linear(...)`: the alphabet adaptation of a fine-tuning run) -- after each, `forward` and an `engine_pool`
    copy serve the NEW weights: equal to a model built from scratch with them."""
    hp = synth.hparams('tiny')
    s0 = synth.make_state_dict(hp, seed=3, decoder_gain=4.0)
    s1 = synth.make_state_dict(hp, seed=4, decoder_gain=4.0)
    image, lens = synth.make_lines(3, hp.height, 64, seed=7, widths=[64, 37, 50])
    x, l = torch.from_numpy(image).cuda(), torch.from_numpy(lens)
    net = _net(hp, s0)
    fresh = lambda st: _net(hp, st).forward(x, l)[0].cpu().numpy()
    cur = {k: torch.from_numpy(np.asarray(v)) for k, v in s0.items()}

    def expect():
        want = fresh({k: v.numpy() for k, v in cur.items()})
        got = net.forward(x, l)[0].cpu().numpy()
        np.testing.assert_array_equal(got, want)
        pool = net.engine_pool(2)
        np.testing.assert_array_equal(pool[1].forward(x[:, 0], lens)[0].cpu().numpy(), want)
        return got

    a = expect()
    with torch.no_grad():                                            # 1. in-place write
        net.nn.decoder.bias.add_(0.5)
    cur['decoder.bias'] = cur['decoder.bias'] + 0.5
    b = expect()
    assert not np.array_equal(a, b)
    key = 'encoder.layers.0.sequential.0.module.sequential.1.linear.weight'
    cur[key] = torch.from_numpy(np.asarray(s1[key]))                 # 2. load_state_dict (copy into the existing tensors)
    net.nn.load_state_dict({k: v.clone() for k, v in cur.items()})
    c = expect()
    assert not np.array_equal(b, c)
    for k in list(cur):                                              # 3. load_state_dict(assign=True): the module's tensors are REPLACED
        if k.endswith('u_bias') or k.endswith('conv.2.weight'):
            cur[k] = torch.from_numpy(np.asarray(s1[k]))
    net.nn.load_state_dict({k: v.clone().cuda() for k, v in cur.items()}, assign=True)
    d = expect()
    assert not np.array_equal(c, d)
    p = net.nn.decoder.weight                                        # 4. p.data = new storage
    cur['decoder.weight'] = cur['decoder.weight'] * 1.25
    p.data = cur['decoder.weight'].clone().cuda()
    e = expect()
    assert not np.array_equal(d, e)
    lin = torch.nn.Linear(hp.encoder_dim, hp.num_classes)            # 5. a new output layer module
    net.nn.decoder = lin.cuda()
    cur['decoder.weight'], cur['decoder.bias'] = lin.weight.detach().cpu().clone(), lin.bias.detach().cpu().clone()
    f = expect()
    assert not np.array_equal(e, f)
    blk = net.nn.encoder.layers._modules['1'].sequential._modules['4']          # 6. a replaced parameter object (block-final LayerNorm gain)
    cur['encoder.layers.1.sequential.4.weight'] = cur['encoder.layers.1.sequential.4.weight'] * 0.5
    blk.weight = torch.nn.Parameter(cur['encoder.layers.1.sequential.4.weight'].clone().cuda(), requires_grad=False)
    g = expect()
    assert not np.array_equal(f, g)
    # and nothing changed: the packed model is kept (no re-pack per call)
    eng = net.engine()
    net.forward(x, l)
    assert net.engine() is eng


def test_fused_argmax_is_not_reused_for_other_values_in_the_same_buffer():
    """The decoder product's epilogue leaves the per-frame argmax for `ctc_greedy`; that shortcut belongs to the tensor object the
    forward returned and to the values it wrote.  (i) another engine writes the same `out` buffer before the decode; (ii) the
    tensor is freed and the allocator hands its address to a new tensor with other logits: both decode from the VALUES."""
    from oracle.ctc_ref import greedy_decoder as ref_greedy
    hp = synth.hparams('cfg2', num_encoder_layers=1)
    sa, sb = synth.make_state_dict(hp, seed=11, decoder_gain=8.0), synth.make_state_dict(hp, seed=12, decoder_gain=8.0)
    A, B = make_engine(hp, sa, 'bf16'), make_engine(hp, sb, 'bf16')
    image, lens = synth.make_lines(4, hp.height, 400, seed=3)
    x = torch.from_numpy(image[:, 0]).cuda()
    T = A.out_len(400)
    buf = torch.empty((4, T, hp.num_classes), dtype=torch.float32, device='cuda')
    la, ol = A.forward(x, lens, out=buf)
    own = [r for r in A.ctc_greedy(la, ol)]
    want_a = [ref_greedy(la[n, :int(ol[n])].cpu().numpy().T) for n in range(4)]
    assert [[r[:3] for r in line] for line in own] == [[r[:3] for r in line] for line in want_a]
    A.forward(x, lens, out=buf)
    lb, _ = B.forward(x, lens, out=buf)                              # (i) a foreign writer into A's buffer
    assert lb is buf
    want_b = [ref_greedy(buf[n, :int(ol[n])].cpu().numpy().T) for n in range(4)]
    assert [[r[:3] for r in line] for line in want_b] != [[r[:3] for r in line] for line in want_a]
    got = A.ctc_greedy(buf, ol)
    assert [[r[:3] for r in line] for line in got] == [[r[:3] for r in line] for line in want_b]
    # (ii) address reuse
    lg, ol = A.forward(x, lens)
    addr = lg.data_ptr()
    del lg
    other = torch.empty((4, T, hp.num_classes), dtype=torch.float32, device='cuda')
    if other.data_ptr() == addr:                                     # the caching allocator returned the freed block
        other.copy_(buf)
        got = A.ctc_greedy(other, ol)
        assert [[r[:3] for r in line] for line in got] == [[r[:3] for r in line] for line in want_b]


def test_predict_string_table_path_equals_the_record_path():
    """`predict_string` maps label arrays to characters by one table lookup per line when the codec is 1:1 (pred.py `_codec_lut`); it must
    return what the reference's loop returns -- the codec's decode over `predict_labels`' records (pred.py:157-164) -- for a codec that
    knows every label, one that misses some (skipped, as `decode` skips them), one whose graphemes are two characters long, and (fallback:
    no table) one with multi-label graphemes; greedy and beam decoders; and through `recognize(streams=2)`."""
    from conformer_ocr_amd.codec import PytorchCodec
    from conformer_ocr_amd.ctc_decoder import BeamDecoder
    from conformer_ocr_amd.evaluate import recognize
    hp = synth.hparams('tiny')
    state = synth.make_state_dict(hp, seed=7, decoder_gain=8.0)
    image, lens = synth.make_lines(3, hp.height, 64, seed=7, widths=[64, 37, 50])
    x, l = torch.from_numpy(image).cuda(), torch.from_numpy(lens)
    net = _net(hp, state)
    full = ascii_codec(hp.num_classes)
    codecs = {
        'full': full,
        'partial': PytorchCodec({k: v for k, v in full.c2l.items() if v[0] % 3 != 0}),
        'digraphs': PytorchCodec({k + k.lower(): v for k, v in full.c2l.items()}),
        'multi-label': PytorchCodec({**{k: v for k, v in full.c2l.items() if v[0] > 2}, 'Q': [1, 2]}),
    }
    seen = set()
    for name, codec in codecs.items():
        net.codec = codec
        for dec in (None, BeamDecoder(4)):
            if dec is not None:
                net.ctc_decoder = dec
            recs = net.predict_labels(x, l)
            want = [''.join(c for c, _, _, _ in codec.decode(r)) for r in recs]
            assert net.predict_string(x, l) == want, (name, dec)
            assert net.collect_strings(net.predict_string_async(x, l)) == want, (name, dec)
            seen.update(want)
        assert (net._codec_lut() is None) == (name == 'multi-label')
    assert len(seen) > 3 and any(seen)
    lines = [image[i, 0, :, :int(lens[i])] for i in range(3)]
    net.codec = codecs['partial']
    assert recognize(net, lines, batch_size=2, streams=2) == recognize(net, lines, batch_size=2, pipelined=False)
