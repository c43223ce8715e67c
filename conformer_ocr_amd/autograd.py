"""The training step behind torch autograd (reference conformer_ocr/model.py:129-152, 283-289).

The reference trains through `loss = self._step(batch)['loss']; loss.backward()` and ANY torch optimizer that
`configure_optimizers` builds over `self.nn.parameters()` (AdamW / Adam / SGD / RMSprop, model.py:283-289), driven by Lightning.
Here the train-mode forward, the CTC criterion and the whole backward run in libcocr_hip.so (`cocr_train_step`); this module is the
thin bridge that makes that step look like one differentiable torch op:

    net = PytorchRecognitionModel(...).to('cuda').train()
    net.nn.requires_grad_(True)
    opt = torch.optim.AdamW(net.nn.parameters(), lr=1e-3, weight_decay=1e-3)        # or SGD, RMSprop, a Lightning module's optimizer ...
    loss = net.training_step(batch)            # 0-dim tensor with a grad_fn            (model.py:147-152)
    loss.backward()                            # .grad on every parameter of net.nn     (autograd accumulates as usual)
    opt.step(); opt.zero_grad()

Per step: the current values of `net.nn`'s parameters and BatchNorm running statistics are copied into the library's flat value vector
(one fused multi-tensor copy, device to device), `cocr_train_step` runs, the flat gradient vector is cloned once and handed to autograd
as per-parameter slices of that clone, and the updated running statistics (+ `num_batches_tracked`) are written back into the module's
buffers, as `nn.BatchNorm1d` in train mode would.  Nothing here evaluates a layer in torch; optimizers, schedulers, DDP and the
Lightning loop stay torch's (out of scope)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .engine import HipRecognizer


def _view(engine: HipRecognizer, ptr: int, n: int) -> torch.Tensor:
    class _Mem:                     # __cuda_array_interface__ view of library-owned memory; `owner` keeps the model alive
        def __init__(s, owner):
            s.owner = owner
            s.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}
    return torch.as_tensor(_Mem(engine), device=engine.device)


class TrainBridge:
    """One training state of the library (`cocr_train_begin`) bound to the parameter holders of a `PytorchRecognitionModel`."""

    def __init__(self, net, matmul_precision: str = 'highest'):
        dev = next(net.nn.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('training runs on a GPU device only (no CPU fallback): move the model to cuda first')
        self.engine = eng = HipRecognizer(net.hparams_record, dev, 'fp32')
        eng.load_state({k: v for k, v in net.nn.state_dict().items()}, strict=True)
        eng.train_begin(matmul_precision)
        self.matmul_precision = matmul_precision
        p, nt, npar = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _lib.check(eng.lib.cocr_train_param_buffer(eng._h, C.byref(p), C.byref(nt), C.byref(npar)))
        self.P = _view(eng, p.value, nt.value)
        self.G = eng.train_grad_buffer()
        self.n_params = int(npar.value)
        self.layout: Dict[str, Tuple[int, int, bool]] = {}
        off, n, isp = C.c_int64(), C.c_int64(), C.c_int()
        for name in net.nn.state_dict().keys():
            if name.endswith('num_batches_tracked'):
                continue
            _lib.check(eng.lib.cocr_train_layout(eng._h, name.encode(), C.byref(off), C.byref(n), C.byref(isp)))
            self.layout[name] = (int(off.value), int(n.value), bool(isp.value))
        self.steps = 0

    def bind(self, net):
        """The module tree's tensors in the layout's order, and the slices of the value vector they are copied to / from.  Rebuilt per
        step from the module's own dicts: a parameter replaced since the last step is simply the one that is read."""
        params = dict(net.nn.named_parameters())
        bufs = dict(net.nn.named_buffers())
        names_p = [k for k, (_, _, isp) in self.layout.items() if isp]
        names_b = [k for k, (_, _, isp) in self.layout.items() if not isp]
        missing = [k for k in names_p if k not in params] + [k for k in names_b if k not in bufs]
        if missing:
            raise RuntimeError('Missing key(s) in the module tree: ' + ', '.join(missing))
        return names_p, [params[k] for k in names_p], names_b, [bufs[k] for k in names_b], bufs

    def slices(self, flat: torch.Tensor, names: List[str], like: List[torch.Tensor]) -> List[torch.Tensor]:
        out = []
        for k, t in zip(names, like):
            off, n, _ = self.layout[k]
            if t.numel() != n:
                raise RuntimeError(f'{k}: the module holds {t.numel()} elements, the model {n}')
            out.append(flat[off:off + n].view(t.shape))
        return out


class _TrainStep(torch.autograd.Function):
    """loss = CTC(train-mode forward(lines)); d loss / d every parameter comes from the same library call."""

    @staticmethod
    def forward(ctx, bridge: TrainBridge, lines, lens, targets, target_lens, dropout, seed, names_p, *params):
        eng = bridge.engine
        with torch.no_grad():
            dst = bridge.slices(bridge.P, names_p, list(params))
            torch._foreach_copy_(dst, [p.detach().to(device=eng.device, dtype=torch.float32) for p in params])
        loss = eng.train_step(lines, lens, targets, target_lens, dropout=dropout, seed=seed)
        grads = bridge.G.clone()                          # this step's gradients, owned by autograd from here on
        ctx.bridge, ctx.names_p = bridge, names_p
        ctx.shapes = [p.shape for p in params]
        ctx.dtypes = [p.dtype for p in params]
        ctx.save_for_backward(grads)
        return torch.tensor(loss, dtype=torch.float32, device=eng.device)

    @staticmethod
    def backward(ctx, grad_loss):
        (grads,) = ctx.saved_tensors
        b = ctx.bridge
        out = []
        for k, shp, dt, need in zip(ctx.names_p, ctx.shapes, ctx.dtypes, ctx.needs_input_grad[8:]):
            if not need:
                out.append(None)
                continue
            off, n, _ = b.layout[k]
            g = grads[off:off + n].view(shp)
            out.append((g * grad_loss).to(dt))
        return (None,) * 8 + tuple(out)


def training_step(net, batch: Dict, seed: int = None) -> torch.Tensor:
    """`RecognitionModel.training_step` (model.py:147-152) for `net` = a `PytorchRecognitionModel` in train mode: returns the batch's
    summed CTC loss as a differentiable 0-dim device tensor.  batch: {'image' (N,1,H,W), 'seq_lens' (N), 'target' (sum target_lens,),
    'target_lens' (N)} -- the reference's batch dict (model.py:131-138)."""
    bridge = getattr(net, '_train_bridge', None)
    dev = next(net.nn.parameters()).device
    if bridge is None or bridge.engine.device != dev or bridge.matmul_precision != getattr(net, 'matmul_precision', 'highest'):
        bridge = net._train_bridge = TrainBridge(net, getattr(net, 'matmul_precision', 'highest'))
    image = batch['image']
    if image.dim() != 4 or image.shape[1] != 1:
        raise ValueError(f'expected a (N,1,H,W) line batch, got {tuple(image.shape)}')
    lines = image.squeeze(1).to(dev)
    names_p, params, names_b, bufs, all_bufs = bridge.bind(net)
    with torch.no_grad():                                 # BatchNorm running statistics as the module holds them now
        if bufs:
            torch._foreach_copy_(bridge.slices(bridge.P, names_b, bufs), [b.detach().to(device=dev, dtype=torch.float32) for b in bufs])
    if seed is None:
        seed = int(getattr(net, 'dropout_seed', 0)) * 1000003 + bridge.steps
    loss = _TrainStep.apply(bridge, lines, torch.as_tensor(batch['seq_lens']).cpu().numpy(), torch.as_tensor(batch['target']).cpu().numpy(),
                            torch.as_tensor(batch['target_lens']).cpu().numpy(), tuple(net.dropout_p), int(seed), names_p, *params)
    with torch.no_grad():                                 # what nn.BatchNorm1d.forward does to its buffers in train mode
        if bufs:
            torch._foreach_copy_(bufs, [s.to(b.dtype) for s, b in zip(bridge.slices(bridge.P, names_b, bufs), bufs)])
        for k, b in all_bufs.items():
            if k.endswith('num_batches_tracked'):
                b.add_(1)
    bridge.steps += 1
    return loss
