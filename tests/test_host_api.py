"""CPU-side checks of the host layer: the C-ABI library loads and exports every symbol include/cocr.h
declares, argument validation mirrors the reference constructors, the codec stand-in, the host class
surface and its loaders.  No compute call is made (no GPU here)."""
import ctypes as C
import io
import json
import os
import re
import tarfile

import numpy as np
import pytest
import torch

from conformer_ocr_amd import _lib, synth
from conformer_ocr_amd.codec import PytorchCodec, ascii_codec
from conformer_ocr_amd.pred import PytorchRecognitionModel, save_safetensors
from conformer_ocr_amd.spec import HParams, model_state_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPS = dict(input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'cocr.h')).read()
    declared = set(re.findall(r'\b(cocr_[a-z_0-9]+)\s*\(', header))
    assert declared, 'no declarations found'
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.cocr_version().startswith(b'cocr-hip')


def test_out_len_entry_point_matches_calc_length():
    lib = _lib.load()
    for w, f, want in [(1200, 4, 300), (512, 4, 128), (300, 4, 75), (37, 4, 10), (96, 8, 12), (1, 4, 1), (2400, 4, 600)]:
        assert lib.cocr_out_len(w, f) == want


@pytest.mark.parametrize('dtype', [np.uint8, np.float32])
@pytest.mark.parametrize('threads', [1, 4])
def test_native_collation_equals_the_python_loop(dtype, threads):
    """cocr_collate_lines = evaluate.collate (left-aligned lines, zero padding), no GPU involved."""
    import ctypes as C
    from conformer_ocr_amd.evaluate import collate
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for N, H, W in [(1, 3, 5), (7, 16, 130), (32, 96, 400)]:
        widths = rng.integers(0 if N > 1 else 1, W + 1, size=N).astype(np.int32)
        widths[0] = W
        lines = [(rng.integers(0, 256, size=(H, w)).astype(dtype) if dtype == np.uint8 else rng.random((H, w), dtype=np.float32))
                 for w in widths]
        dst = np.full((N, H, W), 77, dtype=dtype)
        ptrs = (C.c_void_p * N)(*[a.ctypes.data for a in lines])
        rc = lib.cocr_collate_lines(ptrs, widths.ctypes.data_as(C.POINTER(C.c_int32)), N, H, dst.itemsize, dst.ctypes.data, W, threads)
        assert rc == 0, lib.cocr_last_error()
        want, lens = collate(lines, list(range(N)), W)
        assert np.array_equal(dst, want.numpy()[:, 0].astype(dtype))
        assert lens.tolist() == widths.tolist()
    bad = np.array([9], dtype=np.int32)                       # wider than the batch: refused, nothing written
    one = np.zeros((2, 9), dtype=dtype)
    ptrs = (C.c_void_p * 1)(one.ctypes.data)
    assert lib.cocr_collate_lines(ptrs, bad.ctypes.data_as(C.POINTER(C.c_int32)), 1, 2, one.itemsize, dst.ctypes.data, 8, 1) != 0
    assert b'wide' in lib.cocr_last_error()


def _hp(**kw):
    base = dict(num_classes=11, height=16, encoder_dim=32, num_encoder_layers=1, num_attention_heads=4,
                feed_forward_expansion_factor=4, conv_expansion_factor=2, conv_kernel_size=7, half_step_residual=1,
                subsampling_conv_channels=8, subsampling_factor=4)
    base.update(kw)
    return _lib.HParamsC(**base)


def test_create_validates_like_the_reference_constructors():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.cocr_create(C.byref(_hp()), 0, C.byref(h)) == 0
    buf = C.create_string_buffer(1 << 16)
    n = lib.cocr_missing_tensors(h, buf, len(buf))
    one_layer = HParams(**{**synth.CONFIGS['tiny'], 'num_encoder_layers': 1})
    assert n == len([k for k in model_state_spec(one_layer) if not k.endswith('num_batches_tracked')]) == 48
    lib.cocr_destroy(h)
    for bad, msg in [(dict(num_attention_heads=5), b'num_heads'), (dict(conv_kernel_size=8), b'odd'),
                     (dict(conv_expansion_factor=3), b'expansion_factor 2'), (dict(subsampling_factor=6), b'power of 2')]:
        rc = lib.cocr_create(C.byref(_hp(**bad)), 0, C.byref(h))
        assert rc == _lib.EINVAL and msg in lib.cocr_last_error()
        with pytest.raises(ValueError):
            _lib.check(rc)


def test_set_tensor_rejects_unknown_keys_and_compute_needs_finalize():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.cocr_create(C.byref(_hp()), 0, C.byref(h)) == 0
    a = np.zeros(4, np.float32)
    shape = (C.c_int64 * 1)(4)
    assert lib.cocr_set_tensor(h, b'encoder.nonsense', a.ctypes.data_as(C.c_void_p), _lib.F32, 1, shape) == _lib.EINVAL
    assert lib.cocr_set_tensor(h, b'encoder.layers.0.sequential.2.module.sequential.5.num_batches_tracked',
                               a.ctypes.data_as(C.c_void_p), _lib.I64, 0, shape) == 0
    lens = (C.c_int32 * 1)(40)
    assert lib.cocr_forward(h, C.c_void_p(16), _lib.F32, 1, 16, 40, lens, C.c_void_p(16), lens, None) == _lib.ESTATE
    lib.cocr_destroy(h)


def test_codec_roundtrip_and_longest_match():
    c = PytorchCodec({'a': [1], 'b': [2], 'ch': [3, 4], 'c': [3]})
    assert c.max_label == 4 and len(c) == 4 and c.is_valid
    assert c.encode('abchc') == [1, 2, 3, 4, 3]
    recs = [(1, 0, 1, 0.9), (3, 2, 3, 0.5), (4, 4, 4, 0.7), (3, 6, 6, 0.8), (9, 7, 7, 0.1)]
    assert c.decode(recs) == [('a', 0, 1, 0.9), ('c', 2, 4, 0.7), ('h', 2, 4, 0.7), ('c', 6, 6, 0.8)]   # label 9 is skipped
    assert ''.join(x[0] for x in ascii_codec(11).decode([(1, 0, 0, 1.0), (10, 1, 1, 1.0)])) == '!*'
    with pytest.raises(ValueError):
        PytorchCodec({'a': [0]})


def _net(hp, **kw):
    return PytorchRecognitionModel(**hp.as_dict(), **DROPS, codec=ascii_codec(hp.num_classes), **kw)


def test_host_class_surface_matches_the_reference():
    hp = synth.hparams('tiny')
    net = _net(hp, some_training_only_kwarg=3)             # unknown kwargs are ignored (pred.py:69)
    assert set(net.nn.keys()) == {'encoder', 'decoder'}
    assert (net.height, net.channels, net.width) == (16, 1, 0)
    for name in ('forward', 'predict', 'predict_string', 'predict_labels', 'load_safetensors', 'load_checkpoint'):
        assert callable(getattr(net, name))
    spec = model_state_spec(hp)
    sd = net.nn.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == spec[k][0] for k in spec)
    # no layer of the network is a torch compute module: only parameter holders and the decoder's parameters
    assert not any(isinstance(m, (torch.nn.Conv2d, torch.nn.LayerNorm, torch.nn.BatchNorm1d)) for m in net.modules())
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.forward(torch.zeros(1, 1, 16, 40), torch.tensor([40]))
    with pytest.raises(TypeError):
        net.forward(torch.zeros(1, 1, 16, 40))


def test_safetensors_and_checkpoint_loaders(tmp_path):
    hp = synth.hparams('tiny')
    state = synth.make_state_dict(hp, seed=3)
    net = _net(hp)
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    p = tmp_path / 'model.tar'
    save_safetensors(net, p)
    back = PytorchRecognitionModel.load_safetensors(p)
    assert not back.training and back.codec.c2l == net.codec.c2l
    for k, v in back.nn.state_dict().items():
        assert torch.equal(v, net.nn.state_dict()[k]), k
    # archive without hyper-parameters -> the reference's ValueError (pred.py:190-191)
    bad = tmp_path / 'bad.tar'
    with tarfile.open(bad, 'w') as tf:
        data = json.dumps({'codec': {'a': [1]}}).encode()
        ti = tarfile.TarInfo('metadata.json'); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
    with pytest.raises(ValueError, match='hyperparameters'):
        PytorchRecognitionModel.load_safetensors(bad)
    # lightning-style checkpoint: keys nn.encoder.* / nn.decoder.*, hyper_parameters with training-only extras
    ck = {'state_dict': {'nn.' + k: torch.from_numpy(np.asarray(v)) for k, v in state.items()},
          'hyper_parameters': {**hp.as_dict(), **DROPS, 'lr': 3e-4, 'batch_size': 32},
          'TextLineDataModule': {'codec': {'a': [1], 'b': [2]}}}
    pth = tmp_path / 'ckpt.ckpt'
    torch.save(ck, pth)
    net2 = PytorchRecognitionModel.load_checkpoint(pth)
    assert torch.equal(net2.nn.state_dict()['decoder.weight'], torch.from_numpy(state['decoder.weight']))
    torch.save({'state_dict': {}}, pth)
    with pytest.raises(ValueError, match='data module state'):
        PytorchRecognitionModel.load_checkpoint(pth)
    torch.save({'state_dict': {}, 'TextLineDataModule': {'codec': {'a': [1]}}}, pth)
    with pytest.raises(ValueError, match='No hyperparameters'):
        PytorchRecognitionModel.load_checkpoint(pth)


def test_synthetic_generators_are_deterministic():
    hp = synth.hparams('tiny')
    a, b = synth.make_state_dict(hp, 5), synth.make_state_dict(hp, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a['decoder.weight'], synth.make_state_dict(hp, 6)['decoder.weight'])
    img, lens = synth.make_lines(3, 16, 64, seed=2, widths=[64, 37, 50])
    assert img.shape == (3, 1, 16, 64) and img.dtype == np.float32 and (img[1, :, :, 37:] == 0).all()
    np.testing.assert_array_equal(synth.lines_u8(img).astype(np.float32) / np.float32(255.0), img)


def test_average_checkpoints_like_the_reference(tmp_path):
    """cli/train.py:37-98: mean of the last n checkpoints' state dicts; ints floor-divided; key mismatch raises."""
    import torch
    from conformer_ocr_amd.pred import average_checkpoints
    g = torch.Generator().manual_seed(3)
    paths = []
    for i in range(4):
        sd = {'nn.a.weight': torch.randn(3, 2, generator=g), 'nn.bn.num_batches_tracked': torch.tensor(10 * i + 3),
              'nn.h': torch.randn(4, generator=g).half()}
        f = tmp_path / f'checkpoint_{i:02d}-0.1.ckpt'
        torch.save({'state_dict': sd, 'hyper_parameters': {'height': 96, 'epoch': i}}, f)
        paths.append(f)
    avg = average_checkpoints(paths, 3)
    loaded = [torch.load(p, weights_only=True)['state_dict'] for p in paths[-3:]]
    want = sum(s['nn.a.weight'] for s in loaded) / 3
    assert torch.allclose(avg['state_dict']['nn.a.weight'], want)
    assert int(avg['state_dict']['nn.bn.num_batches_tracked']) == (13 + 23 + 33) // 3
    assert avg['state_dict']['nn.h'].dtype == torch.float32
    assert avg['hyper_parameters']['epoch'] == 1            # everything else from the first averaged checkpoint
    with pytest.raises(ValueError):
        average_checkpoints(paths, 9)
    bad = tmp_path / 'checkpoint_99-0.1.ckpt'
    torch.save({'state_dict': {'other': torch.zeros(1)}}, bad)
    with pytest.raises(KeyError):
        average_checkpoints(paths + [bad], 2)


def test_alignment_report_known_answers():
    """The report of the reference's test loop (cli/test.py:194-224): alignment, confusion tallies, rendered text.  kraken's own
    functions are absent (parity unpinned): hand-derived cases."""
    from conformer_ocr_amd.evaluate import compute_confusions, edit_distance, global_align, render_report
    c, a, b = global_align('kitten', 'sitting')
    assert c == 3 == edit_distance('kitten', 'sitting') and len(a) == len(b) == 7
    assert ''.join(a) == 'kitten' and ''.join(b) == 'sitting' and a[-1] == '' and b[-1] == 'g'
    c, a, b = global_align('abc', 'abc')
    assert c == 0 and a == b == list('abc')
    c, a, b = global_align('', 'xy')
    assert c == 2 and a == ['', ''] and b == ['x', 'y']
    conf, scripts, ins, dels, subs = compute_confusions(list('kitten') + [''], list('sittin') + ['g'])
    assert conf == {('k', 's'): 1, ('e', 'i'): 1, ('', 'g'): 1} and scripts == {'Latin': 6} and ins == {'Latin': 1} and dels == 0 and subs == {'Latin': 2}
    conf, scripts, ins, dels, subs = compute_confusions(['a', 'b', '1'], ['a', '', '7'])
    assert conf == {('b', ''): 1, ('1', '7'): 1} and dels == 1 and subs == {'Digit': 1} and scripts == {'Latin': 2, 'Digit': 1}
    rep = render_report('m', 6, 3, 0.5, 0.0, *compute_confusions(list('kitten') + [''], list('sittin') + ['g']))
    assert '6\tCharacters' in rep and '3\tErrors' in rep and '50.00%\tCharacter Accuracy' in rep and '{ k } - { s }' in rep


def test_signature_sees_every_kind_of_weight_change():
    """pred.py `_signature` (CPU: no packed model is built): in-place writes, load_state_dict with and without assign=True, `p.data =`,
    a replaced parameter, a replaced sub-module, two parameters swapped between modules -- each changes it; nothing changes it otherwise."""
    hp = synth.hparams('tiny')
    net = _net(hp)
    dev = torch.device('cpu')
    seen = [net._signature(dev)]

    def changed():
        s = net._signature(dev)
        assert s == net._signature(dev)
        assert all(s != t for t in seen), 'a weight change went unnoticed'
        seen.append(s)

    assert net._signature(dev) == seen[0]
    with torch.no_grad():
        net.nn.decoder.bias.add_(1.0)
    changed()
    net.nn.load_state_dict({k: v.clone() for k, v in net.nn.state_dict().items()})
    changed()
    net.nn.load_state_dict({k: v.clone() for k, v in net.nn.state_dict().items()}, assign=True)
    changed()
    net.nn.decoder.weight.data = torch.zeros_like(net.nn.decoder.weight)
    changed()
    net.nn.decoder = torch.nn.Linear(hp.encoder_dim, hp.num_classes)
    changed()
    a = net.nn.encoder.layers._modules['0'].sequential._modules['4']
    b = net.nn.encoder.layers._modules['1'].sequential._modules['4']
    a.weight, b.weight = b.weight, a.weight
    changed()
    a.weight = torch.nn.Parameter(torch.ones_like(a.weight), requires_grad=False)
    changed()
    net.float()
    assert net._signature(dev) == net._signature(dev)


def test_training_step_has_no_cpu_fallback():
    """`net.training_step` (the step behind torch autograd) needs the GPU library like every other compute call; frozen holders say what to do."""
    hp = synth.hparams('tiny')
    net = _net(hp)
    batch = {'image': torch.zeros(1, 1, hp.height, 40), 'seq_lens': torch.tensor([40]), 'target': torch.tensor([1, 2]), 'target_lens': torch.tensor([2])}
    with pytest.raises(RuntimeError, match='requires_grad_'):
        net.training_step(batch)
    net.nn.requires_grad_(True)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.training_step(batch)
