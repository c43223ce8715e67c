"""GPU: the output-layer training step (decoder backward + AdamW) against torch autograd / torch.optim.AdamW on the CPU -- the
implementations the reference's `training_step` and `configure_optimizers` use (model.py:147-152,283-284)."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.codec import ascii_codec
from conformer_ocr_amd.pred import PytorchRecognitionModel
from conformer_ocr_amd.train import DecoderTrainer
from tests.hip_util import make_engine

pytestmark = pytest.mark.gpu


def _batch(hp, N, W, seed):
    img, lens = synth.make_lines(N, hp.height, W, seed=seed)
    g = np.random.default_rng(seed)
    label_lens = g.integers(1, 10, size=N)
    target = np.concatenate([g.integers(1, hp.num_classes, size=l) for l in label_lens])
    return img, lens, target, label_lens


def _torch_reference(y, W, b, target, out_lens, label_lens):
    """The reference's lines: decoder linear -> log_softmax -> CTCLoss(sum, zero_infinity) -> backward (model.py:135-142)."""
    y = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    W = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    probits = y @ W.T + b
    logits = torch.nn.functional.log_softmax(probits, dim=-1)
    loss = torch.nn.CTCLoss(reduction='sum', zero_infinity=True)(logits.transpose(0, 1), torch.tensor(target), torch.tensor(out_lens), torch.tensor(label_lens))
    loss.backward()
    return loss.item(), W.grad.numpy(), b.grad.numpy(), y.grad.numpy()


@pytest.mark.parametrize('name,dtype', [('cfg1', 'fp32'), ('cfg1', 'bf16')])
def test_decoder_backward_matches_autograd(case, name, dtype):
    hp, state, *_ = case(name)
    img, lens, target, label_lens = _batch(hp, 5, 264, 11)
    eng = make_engine(hp, state, dtype)
    eng.set_debug(True)         # keeps a float32 copy of the encoder output (tap "l<last>.out"); the operand the decoder multiplies is its cast
    probits, out_lens = eng.forward(torch.from_numpy(img[:, 0]).cuda(), lens)
    y = eng.tap(f'l{hp.num_encoder_layers - 1}.out').reshape(probits.shape[0], probits.shape[1], hp.encoder_dim)
    nll, grad = eng.ctc_loss(probits, out_lens, target, label_lens)
    gw, gb, gy = eng.decoder_backward(grad, with_input_grad=True)
    gw2, gb2, _ = eng.decoder_backward(grad)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)              # reproducible bit for bit (fixed-order reductions)
    with pytest.raises(RuntimeError):
        eng.decoder_backward(grad[:2])                                 # the gradient must belong to the last forward
    torch.cuda.synchronize()
    W, b = np.asarray(state['decoder.weight'], np.float32), np.asarray(state['decoder.bias'], np.float32)
    if dtype == 'bf16':      # the forward multiplied bf16(y) with bf16(W)
        y = torch.from_numpy(y).bfloat16().float().numpy()
        W = torch.from_numpy(W).bfloat16().float().numpy()
    loss, want_gw, want_gb, want_gy = _torch_reference(y, W, b, target, out_lens, label_lens)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == 'fp32' else dict(rtol=2e-3, atol=5e-3)
    assert abs(float(nll.sum()) - loss) < 1e-4 * loss
    np.testing.assert_allclose(gw.cpu().numpy(), want_gw, **tol)
    np.testing.assert_allclose(gb.cpu().numpy(), want_gb, **tol)
    np.testing.assert_allclose(gy.cpu().numpy(), want_gy, **tol)


def test_adamw_matches_torch(case):
    hp, state, *_ = case('cfg1')
    eng = make_engine(hp, state, 'fp32')
    W0, b0 = np.asarray(state['decoder.weight'], np.float32), np.asarray(state['decoder.bias'], np.float32)
    Wt, bt = torch.nn.Parameter(torch.from_numpy(W0.copy())), torch.nn.Parameter(torch.from_numpy(b0.copy()))
    opt = torch.optim.AdamW([Wt, bt], lr=3e-3, weight_decay=1e-2, betas=(0.9, 0.98), eps=1e-7)
    g = torch.Generator().manual_seed(0)
    for step in range(4):
        gw = torch.randn(W0.shape, generator=g) * (10.0 ** (step - 2))
        gb = torch.randn(b0.shape, generator=g)
        Wt.grad, bt.grad = gw.clone(), gb.clone()
        opt.step()
        eng.decoder_adamw(gw.cuda(), gb.cuda(), 3e-3, (0.9, 0.98), 1e-7, 1e-2)
        st = eng.decoder_state()
        np.testing.assert_allclose(st['decoder.weight'], Wt.detach().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(st['decoder.bias'], bt.detach().numpy(), rtol=2e-6, atol=2e-7)
    with pytest.raises(ValueError):
        eng.decoder_adamw(gw.cuda(), gb.cuda(), 1e-3, (1.5, 0.9))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_training_lowers_the_loss_and_the_forward_sees_the_update(case, dtype):
    hp, state, *_ = case('cfg1')
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1,
                                  codec=ascii_codec(hp.num_classes), compute_dtype=dtype)
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0').eval()
    img, lens, target, label_lens = _batch(hp, 6, 264, 23)
    batch = {'image': torch.from_numpy(img).cuda(), 'seq_lens': torch.from_numpy(lens), 'target': torch.from_numpy(target), 'target_lens': torch.from_numpy(label_lens)}
    tr = DecoderTrainer(net, lrate=2e-2, weight_decay=0.0)
    losses = [float(tr.training_step(batch)) for _ in range(12)]
    assert losses[-1] < 0.6 * losses[0], losses
    assert all(b <= a * 1.05 for a, b in zip(losses, losses[1:])), losses
    after = float(net.step(batch)['loss'])
    assert after < losses[-1] * 1.02
    # module <- engine, engine (and its optimizer state) kept
    eng = net._engine
    w_before = net.nn['decoder'].weight.detach().clone()
    tr.sync_module()
    assert not torch.equal(w_before, net.nn['decoder'].weight)
    assert net.engine() is eng
    assert abs(float(net.step(batch)['loss']) - after) < 1e-3 * after
    # a fresh engine packed from the synced module computes the same loss (bf16: the master copy is rounded again, identically)
    net._engine = None
    assert abs(float(net.step(batch)['loss']) - after) < 2e-3 * after
