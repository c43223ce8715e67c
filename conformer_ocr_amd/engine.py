"""Thin Python handle over a `cocr_model` (include/cocr.h).  torch is used only to own device
memory and streams; every computation is a call through the C ABI."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .spec import HParams

DTYPES = {'bf16': _lib.BF16, 'bfloat16': _lib.BF16, 'fp32': _lib.F32, 'f32': _lib.F32, 'float32': _lib.F32}


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_LAST_WRITER: Dict[int, int] = {}          # logits buffer address -> sequence number of the last `forward` (of any engine) that wrote it


class HipRecognizer:
    """One packed model on one GPU."""
    _WRITE_SEQ = 0

    def __init__(self, hp: HParams, device: torch.device, compute_dtype: str = 'bf16'):
        if compute_dtype not in DTYPES:
            raise ValueError(f'compute_dtype must be one of {sorted(DTYPES)}')
        self.lib = _lib.load()
        self.hp = hp
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('the HIP recognizer runs on a GPU device only (no CPU fallback)')
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device('cuda', self.dev_index)
        self.compute_dtype = compute_dtype
        chp = _lib.HParamsC(num_classes=hp.num_classes, height=hp.height, encoder_dim=hp.encoder_dim,
                            num_encoder_layers=hp.num_encoder_layers, num_attention_heads=hp.num_attention_heads,
                            feed_forward_expansion_factor=hp.feed_forward_expansion_factor,
                            conv_expansion_factor=hp.conv_expansion_factor, conv_kernel_size=hp.conv_kernel_size,
                            half_step_residual=int(bool(hp.half_step_residual)),
                            subsampling_conv_channels=hp.subsampling_conv_channels,
                            subsampling_factor=hp.subsampling_factor)
        h = C.c_void_p()
        _lib.check(self.lib.cocr_create(C.byref(chp), self.dev_index, C.byref(h)))
        self._h = h
        self.ready = False
        self._pinned = {}

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                self.lib.cocr_destroy(h)
            except Exception:
                pass

    # ---- weights -------------------------------------------------------------------------------
    def load_state(self, state: Mapping[str, 'np.ndarray | torch.Tensor'], strict: bool = True) -> None:
        """`state`: keys of `nn.state_dict()` (`encoder.*`, `decoder.*`)."""
        for name, val in state.items():
            if name.endswith('num_batches_tracked'):
                continue
            arr = val.detach().cpu().float().numpy() if isinstance(val, torch.Tensor) else np.asarray(val, dtype=np.float32)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self.lib.cocr_set_tensor(self._h, name.encode(), arr.ctypes.data_as(C.c_void_p), _lib.F32, arr.ndim, shape))
        if strict:
            buf = C.create_string_buffer(1 << 16)
            n = self.lib.cocr_missing_tensors(self._h, buf, len(buf))
            if n:
                raise RuntimeError('Missing key(s) in state_dict: ' + ', '.join(buf.value.decode().split()))

    def finalize(self) -> None:
        _lib.check(self.lib.cocr_finalize(self._h, DTYPES[self.compute_dtype]))
        self.ready = True

    def share_weights(self, owner: 'HipRecognizer') -> None:
        """This packed copy reads `owner`'s weights and tables instead of its own (include/cocr.h: cocr_share_weights); keeps `owner` alive."""
        _lib.check(self.lib.cocr_share_weights(self._h, owner._h))
        self._weights_owner = owner
        self.ready = True

    def finalize_empty(self) -> None:
        _lib.check(self.lib.cocr_finalize_empty(self._h, DTYPES[self.compute_dtype]))
        self.ready = True

    def weight_blob(self) -> torch.Tensor:
        """The packed device blob as a uint8 torch view (for an RCCL broadcast through torch.distributed)."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.lib.cocr_weight_blob(self._h, C.byref(p), C.byref(n)))

        class _Mem:   # __cuda_array_interface__ view of library-owned memory; `owner` keeps the model alive
            def __init__(s, owner):
                s.owner = owner
                s.__cuda_array_interface__ = {'shape': (n.value,), 'typestr': '|u1', 'data': (p.value, False), 'version': 2}
        return torch.as_tensor(_Mem(self), device=self.device)

    def blob_nbytes(self) -> int:
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.lib.cocr_weight_blob(self._h, C.byref(p), C.byref(n)))
        return int(n.value)

    def export_blob(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Packed weights copied into a torch-owned uint8 tensor (the buffer a collective runs on)."""
        n = self.blob_nbytes()
        buf = out if out is not None else torch.empty(n, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_blob_export(self._h, C.c_void_p(buf.data_ptr()), n, _stream_ptr(self.device)))
        return buf

    def import_blob(self, buf: torch.Tensor) -> None:
        if buf.dtype != torch.uint8 or buf.device != self.device or not buf.is_contiguous():
            raise ValueError('blob buffer must be a contiguous uint8 tensor on the model device')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_blob_import(self._h, C.c_void_p(buf.data_ptr()), buf.numel(), _stream_ptr(self.device)))

    # ---- compute -------------------------------------------------------------------------------
    def out_len(self, w: int) -> int:
        return int(self.lib.cocr_out_len(int(w), self.hp.subsampling_factor))

    def reserve(self, n: int, w: int) -> None:
        _lib.check(self.lib.cocr_reserve(self._h, int(n), int(w)))

    def set_graph(self, on: bool) -> None:
        """hipGraph replay of repeated identical forwards (same input / output buffers): see include/cocr.h."""
        _lib.check(self.lib.cocr_set_graph(self._h, int(on)))

    def set_chain_rows(self, rows: int) -> None:
        """Rows per workgroup of the row-chain kernels (0 = automatic): see include/cocr.h."""
        _lib.check(self.lib.cocr_set_chain_rows(self._h, int(rows)))

    def forward(self, lines: torch.Tensor, lens: Sequence[int], out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, np.ndarray]:
        """lines: (N,H,W) float32 or uint8 on this device, contiguous.  Returns (logits (N,T,ncls) f32 device, out_lens int32 host).
        `out`: optional preallocated logits buffer (keeps the output address fixed for graph replay)."""
        if not self.ready:
            raise RuntimeError('model not finalized')
        if lines.device != self.device:
            raise RuntimeError(f'line batch lives on {lines.device}, the model on {self.device}')
        if lines.dim() != 3:
            raise ValueError('expected a (N,H,W) batch')
        if lines.dtype == torch.uint8:
            ldt = _lib.U8
        else:
            lines = lines.float()
            ldt = _lib.F32
        lines = lines.contiguous()
        N, H, W = lines.shape
        in_lens = np.ascontiguousarray(np.asarray(lens, dtype=np.int32).reshape(-1))
        if in_lens.shape[0] != N:
            raise ValueError('lens must have one entry per line')
        out_lens = np.zeros(N, dtype=np.int32)
        T = self.out_len(W)
        logits = out if out is not None else torch.empty((N, T, self.hp.num_classes), dtype=torch.float32, device=self.device)
        if logits.shape != (N, T, self.hp.num_classes) or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError('out must be a contiguous float32 (N,T,num_classes) tensor')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_forward(self._h, C.c_void_p(lines.data_ptr()), ldt, N, H, W,
                                             in_lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(logits.data_ptr()),
                                             out_lens.ctypes.data_as(C.POINTER(C.c_int32)), _stream_ptr(self.device)))
        # The decoder epilogue's per-frame argmax belongs to THESE values: to this tensor object (not to its address: the caching
        # allocator hands a freed address to the next same-sized tensor), at this version, and only while no other engine's forward
        # has written the same buffer since (writes through the C ABI do not bump `_version`: `_LAST_WRITER` records them).
        HipRecognizer._WRITE_SEQ += 1
        if len(_LAST_WRITER) > 4096:
            _LAST_WRITER.clear()
        _LAST_WRITER[logits.data_ptr()] = HipRecognizer._WRITE_SEQ
        self._fresh_logits = (weakref.ref(logits), logits._version, HipRecognizer._WRITE_SEQ)
        return logits, out_lens

    def _argmax_is_fresh(self, logits: torch.Tensor) -> bool:
        ref, ver, seq = getattr(self, '_fresh_logits', None) or (None, -1, -1)
        return ref is not None and ref() is logits and logits._version == ver and _LAST_WRITER.get(logits.data_ptr()) == seq

    def _decode_async(self, fn, logits: torch.Tensor, out_lens, extra=()):
        """Enqueues the decode kernel on the current stream, its outputs in pinned host memory; returns a handle for `collect`
        (the pinned buffers + an event)."""
        if logits.device != self.device or logits.dtype != torch.float32:
            raise RuntimeError('logits must be float32 on the model device')
        fresh = self._argmax_is_fresh(logits)
        logits = logits.contiguous()
        N, T, ncls = logits.shape
        lens = np.ascontiguousarray(np.asarray(out_lens, dtype=np.int32).reshape(-1))
        if not fresh:
            _lib.check(self.lib.cocr_forget_argmax(self._h))          # other logits, or edited in place since the forward: decode from the values
        with torch.cuda.device(self.device):
            # The decode kernels write their (sparse) label records STRAIGHT into pinned host memory (device-visible): no
            # device->host copy is enqueued.  A third hipMemcpyAsync D2H in flight (three batches on three streams, each queued
            # behind its forward) blocked the host for a whole forward, 5.5 ms at every start of a run.
            key = (N, T)
            pool = self._pinned.setdefault(key, [])
            host = pool.pop() if pool else (torch.empty((3, N, T), dtype=torch.int32).pin_memory(),
                                            torch.empty((N, T), dtype=torch.float32).pin_memory(),
                                            torch.empty((N,), dtype=torch.int32).pin_memory())
            ints, conf, counts = host
            _lib.check(fn(self._h, C.c_void_p(logits.data_ptr()), N, T, ncls, lens.ctypes.data_as(C.POINTER(C.c_int32)),
                          C.c_void_p(ints[0].data_ptr()), C.c_void_p(ints[1].data_ptr()), C.c_void_p(ints[2].data_ptr()),
                          C.c_void_p(conf.data_ptr()), C.c_void_p(counts.data_ptr()), T, *extra, _stream_ptr(self.device)))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        return (key, host, ev, (logits,))

    def collect(self, handle) -> List[List[Tuple[int, int, int, float]]]:
        """Waits for a `_decode_async` handle and builds the per-line (label, start, end, conf) lists."""
        key, host, ev, _keep = handle
        ev.synchronize()
        ints_h, conf_h, cnt = host[0].numpy(), host[1].numpy(), host[2].numpy()
        # one conversion per array for the whole batch (per-line slices + tolist: 110 us for 32 lines, after the GPU has finished)
        cl = cnt[:key[0]].tolist()
        mx = max(cl) if cl else 0
        lab, st, en = ints_h[:, :, :mx].tolist()
        cf = conf_h[:, :mx].tolist()
        out = [list(zip(lab[n][:c], st[n][:c], en[n][:c], cf[n][:c])) for n, c in enumerate(cl)]
        self._pinned[key].append(host)
        return out

    def collect_labels(self, handle) -> List[np.ndarray]:
        """Waits for a `_decode_async` handle; per line the int32 label array only (copies: the pinned buffers go back to the pool) --
        what `predict_string` needs, without building four Python lists per line."""
        key, host, ev, _keep = handle
        ev.synchronize()
        lab, cnt = host[0].numpy()[0], host[2].numpy()
        cl = cnt[:key[0]].tolist()
        mx = max(cl) if cl else 0
        block = lab[:, :mx].copy()
        self._pinned[key].append(host)
        return [block[n, :c] for n, c in enumerate(cl)]

    def _decode(self, fn, logits: torch.Tensor, out_lens, extra=()) -> List[List[Tuple[int, int, int, float]]]:
        return self.collect(self._decode_async(fn, logits, out_lens, extra))

    def ctc_greedy_async(self, logits: torch.Tensor, out_lens):
        return self._decode_async(self.lib.cocr_ctc_greedy, logits, out_lens)

    def ctc_greedy(self, logits: torch.Tensor, out_lens) -> List[List[Tuple[int, int, int, float]]]:
        return self._decode(self.lib.cocr_ctc_greedy, logits, out_lens)

    def ctc_beam(self, logits: torch.Tensor, out_lens, beam: int = 16) -> List[List[Tuple[int, int, int, float]]]:
        return self._decode(self.lib.cocr_ctc_beam, logits, out_lens, extra=(int(beam),))

    def ctc_loss(self, probits: torch.Tensor, out_lens, targets, label_lens, with_grad: bool = True):
        """nn.CTCLoss(reduction='sum', zero_infinity=True) on log_softmax(probits) (reference model.py:119,136-142) on the device.
        probits (N,T,ncls) float32 on this device; out_lens (N) valid frames; targets: the batch's concatenated 1-D label vector,
        label_lens (N) its per-line lengths (host sequences).  Returns (per-line nll (N) float32 device tensor -- 0 where no
        alignment exists --, d sum(nll) / d probits (N,T,ncls) or None); the reference's loss is `nll.sum()`.  Stream-ordered."""
        if probits.device != self.device or probits.dtype != torch.float32 or probits.dim() != 3:
            raise RuntimeError('probits must be a float32 (N,T,num_classes) tensor on the model device')
        probits = probits.contiguous()
        N, T, ncls = probits.shape
        lens = np.ascontiguousarray(np.asarray(out_lens, dtype=np.int32).reshape(-1))
        tl = np.ascontiguousarray(np.asarray(label_lens, dtype=np.int32).reshape(-1))
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32).reshape(-1))
        if lens.shape[0] != N or tl.shape[0] != N:
            raise ValueError('out_lens and label_lens need one entry per line')
        if int(tl.sum()) != tg.shape[0]:
            raise ValueError('targets must hold sum(label_lens) labels')
        nll = torch.empty((N,), dtype=torch.float32, device=self.device)
        grad = torch.empty_like(probits) if with_grad else None
        i32 = C.POINTER(C.c_int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_ctc_loss(self._h, C.c_void_p(probits.data_ptr()), N, T, ncls, lens.ctypes.data_as(i32),
                                              tg.ctypes.data_as(i32) if tg.shape[0] else None, tl.ctypes.data_as(i32),
                                              C.c_void_p(nll.data_ptr()), C.c_void_p(grad.data_ptr()) if with_grad else None,
                                              _stream_ptr(self.device)))
        return nll, grad

    # ---- output-layer training step (include/cocr.h: cocr_decoder_backward / cocr_decoder_adamw) ----------
    def decoder_backward(self, grad_probits: torch.Tensor, with_input_grad: bool = False):
        """Gradients of the decoder `nn.Linear` for the LAST forward on this engine: (grad_weight (ncls, D), grad_bias (ncls),
        grad_output (N, T, D) or None), float32 device tensors."""
        if grad_probits.device != self.device or grad_probits.dtype != torch.float32 or grad_probits.dim() != 3:
            raise RuntimeError('grad_probits must be a float32 (N,T,num_classes) tensor on the model device')
        grad_probits = grad_probits.contiguous()
        N, T, ncls = grad_probits.shape
        if ncls != self.hp.num_classes:
            raise ValueError('grad_probits has the wrong number of classes')
        D = self.hp.encoder_dim
        gw = torch.empty((ncls, D), dtype=torch.float32, device=self.device)
        gb = torch.empty((ncls,), dtype=torch.float32, device=self.device)
        gy = torch.empty((N, T, D), dtype=torch.float32, device=self.device) if with_input_grad else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_decoder_backward(self._h, C.c_void_p(grad_probits.data_ptr()), N, T, C.c_void_p(gw.data_ptr()),
                                                      C.c_void_p(gb.data_ptr()), C.c_void_p(gy.data_ptr()) if with_input_grad else None,
                                                      _stream_ptr(self.device)))
        return gw, gb, gy

    def decoder_adamw(self, grad_weight: torch.Tensor, grad_bias: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                      weight_decay: float = 1e-2) -> None:
        """One torch.optim.AdamW step on the decoder (state kept inside the engine); defaults are torch's."""
        for g in (grad_weight, grad_bias):
            if g.device != self.device or g.dtype != torch.float32 or not g.is_contiguous():
                raise RuntimeError('gradients must be contiguous float32 tensors on the model device')
        if grad_weight.shape != (self.hp.num_classes, self.hp.encoder_dim) or grad_bias.shape != (self.hp.num_classes,):
            raise ValueError('gradient shapes do not match the decoder')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_decoder_adamw(self._h, C.c_void_p(grad_weight.data_ptr()), C.c_void_p(grad_bias.data_ptr()), float(lr),
                                                   float(betas[0]), float(betas[1]), float(eps), float(weight_decay), _stream_ptr(self.device)))

    def decoder_state(self) -> Dict[str, np.ndarray]:
        """{'decoder.weight', 'decoder.bias'}: float32 host copies of the (trained) output layer."""
        out = {}
        for name, shape in (('decoder.weight', (self.hp.num_classes, self.hp.encoder_dim)), ('decoder.bias', (self.hp.num_classes,))):
            a = np.empty(shape, dtype=np.float32)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.cocr_get_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size, _stream_ptr(self.device)))
            out[name] = a
        return out

    # ---- training step of the whole network (include/cocr.h: cocr_train_*) ------------------------------
    def train_begin(self, matmul_precision: str = 'highest') -> None:
        """fp32 master copy of the loaded state (`load_state`) on the device, zeroed AdamW moments.  matmul_precision: 'highest' (exact
        fp32 products) or 'medium' (bf16-rounded operands, fp32 accumulation: torch.set_float32_matmul_precision('medium'), what the
        reference's cli/train.py:252 sets)."""
        if matmul_precision not in ('highest', 'medium'):
            raise ValueError("matmul_precision must be 'highest' or 'medium'")
        _lib.check(self.lib.cocr_train_begin(self._h))
        _lib.check(self.lib.cocr_train_set_matmul(self._h, int(matmul_precision == 'medium')))

    def train_step(self, lines: torch.Tensor, lens, targets, label_lens, dropout=(0.0, 0.0, 0.0, 0.0), seed: int = 0) -> float:
        """`RecognitionModel.training_step` without the optimizer (reference model.py:129-152): train-mode forward, summed CTC loss,
        backward through decoder and encoder.  lines (N,H,W) float32 / uint8 on this device; lens pixel widths; targets the
        concatenated labels, label_lens their per-line counts; dropout = (input, feed_forward, attention, conv) probabilities.
        Returns the loss; the gradients stay on the device (`train_grad`, `train_adamw`)."""
        if lines.device != self.device or lines.dim() != 3:
            raise ValueError('expected a (N,H,W) batch on the model device')
        ldt = _lib.U8 if lines.dtype == torch.uint8 else _lib.F32
        lines = (lines if ldt == _lib.U8 else lines.float()).contiguous()
        N, H, W = lines.shape
        i32 = C.POINTER(C.c_int32)
        il = np.ascontiguousarray(np.asarray(lens, dtype=np.int32).reshape(-1))
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32).reshape(-1))
        tl = np.ascontiguousarray(np.asarray(label_lens, dtype=np.int32).reshape(-1))
        if il.shape[0] != N or tl.shape[0] != N or int(tl.sum()) != tg.shape[0]:
            raise ValueError('lens / label_lens need one entry per line and targets sum(label_lens) labels')
        dp = (C.c_float * 4)(*[float(x) for x in dropout])
        loss = C.c_float()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_train_step(self._h, C.c_void_p(lines.data_ptr()), ldt, N, H, W, il.ctypes.data_as(i32),
                                                tg.ctypes.data_as(i32) if tg.shape[0] else None, tl.ctypes.data_as(i32), dp, C.c_uint64(int(seed)),
                                                C.byref(loss), _stream_ptr(self.device)))
        return float(loss.value)

    def _train_get(self, name: str, kind: int) -> np.ndarray:
        from .spec import model_state_spec
        shape = model_state_spec(self.hp)[name][0]
        a = np.empty(shape, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_train_get(self._h, name.encode(), kind, a.ctypes.data_as(C.c_void_p), a.size, _stream_ptr(self.device)))
        return a

    def train_grad(self, name: str) -> np.ndarray:
        """d loss / d `name` of the last `train_step` (reference state-dict name, e.g. 'encoder.layers.0.sequential.1.module.attention.u_bias')."""
        return self._train_get(name, 1)

    def train_value(self, name: str) -> np.ndarray:
        """Current value of a parameter or buffer (BatchNorm running statistics) of the training state."""
        return self._train_get(name, 0)

    def train_adamw(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
        """One torch.optim.AdamW step on all parameters with the gradients of the last `train_step` (defaults are torch's)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cocr_train_adamw(self._h, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), _stream_ptr(self.device)))

    def train_grad_buffer(self) -> torch.Tensor:
        """The flat gradient vector of all parameters as a float32 torch view of library-owned memory (for an all-reduce)."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.lib.cocr_train_grad_buffer(self._h, C.byref(p), C.byref(n)))

        class _Mem:
            def __init__(s, owner):
                s.owner = owner
                s.__cuda_array_interface__ = {'shape': (n.value,), 'typestr': '<f4', 'data': (p.value, False), 'version': 2}
        return torch.as_tensor(_Mem(self), device=self.device)

    def train_end(self) -> None:
        """Trained values back into the model's state; `finalize()` again to serve them."""
        _lib.check(self.lib.cocr_train_end(self._h))

    # ---- line pre-processing (include/cocr.h: cocr_preproc_lines) --------------------------------------
    def preprocess(self, lines: Sequence[np.ndarray], height: Optional[int] = None, pad: int = 16, width: int = 0,
                   bucket_edge: int = 0) -> Tuple[torch.Tensor, np.ndarray]:
        """Raw 8-bit line crops ((H, W) grayscale or (H, W, 3) RGB numpy arrays, any height) -> the (N, height, Wb) uint8 device
        batch `forward` ingests, and the lines' widths after scaling and padding (the batch's `seq_lens`).  Grayscale, Pillow
        LANCZOS scaling to `height` (default: the model's), `pad` zero columns left and right, right-zero-filled to Wb = `width`,
        or the widest line rounded up to a multiple of `bucket_edge`, or the widest line."""
        height = int(height or self.hp.height)
        n = len(lines)
        if n < 1:
            raise ValueError('empty batch')
        hs = np.array([x.shape[0] for x in lines], dtype=np.int32)
        ws = np.array([x.shape[1] for x in lines], dtype=np.int32)
        ch = np.array([1 if x.ndim == 2 else x.shape[2] for x in lines], dtype=np.int32)
        for x in lines:
            if x.dtype != np.uint8 or x.ndim not in (2, 3):
                raise ValueError('line crops are (H, W) or (H, W, 3) uint8 arrays')
        sizes = hs.astype(np.int64) * ws * ch
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(x).reshape(-1) for x in lines])
        wl = [int(self.lib.cocr_preproc_width(int(h), int(w), height, int(pad))) for h, w in zip(hs, ws)]
        wb = int(width) if width else max(wl)
        if not width and bucket_edge:
            wb = -(-wb // int(bucket_edge)) * int(bucket_edge)
        px = torch.from_numpy(flat).to(self.device)
        out = torch.empty((n, height, wb), dtype=torch.uint8, device=self.device)
        lens = np.zeros(n, dtype=np.int32)
        _lib.check(self.lib.cocr_preproc_lines(self._h, C.c_void_p(px.data_ptr()), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                               hs.ctypes.data_as(C.POINTER(C.c_int32)), ws.ctypes.data_as(C.POINTER(C.c_int32)),
                                               ch.ctypes.data_as(C.POINTER(C.c_int32)), n, height, int(pad), wb,
                                               C.c_void_p(out.data_ptr()), lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                               _stream_ptr(self.device)))
        self._keep = px            # the kernels read it asynchronously on the current stream
        return out, lens

    # ---- test / measurement hooks ---------------------------------------------------------------
    def set_debug(self, on: bool) -> None:
        _lib.check(self.lib.cocr_set_debug(self._h, int(on)))

    def tap(self, name: str) -> np.ndarray:
        n = C.c_int64()
        _lib.check(self.lib.cocr_debug_tap(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        _lib.check(self.lib.cocr_debug_tap(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def profile(self, on: bool) -> None:
        _lib.check(self.lib.cocr_profile(self._h, int(on)))

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)()
        cnt = (C.c_int64 * 64)()
        n = _lib.check(self.lib.cocr_profile_read(self._h, names, len(names), ms, cnt, 64))
        keys = names.value.decode().split('\n')[:n]
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(keys)}
