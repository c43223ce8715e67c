"""
conformer_ocr_amd.pred
~~~~~~~~~~~~~~~~~~~~~~

Recognition inference: `PytorchRecognitionModel` with the surface of the reference class
(reference conformer_ocr/pred.py:50-210 -- constructor arguments, attributes `nn`, `codec`,
`ctc_decoder`, `height`, `channels`, `width`; methods `forward`, `predict`, `predict_string`,
`predict_labels`, `load_safetensors`, `load_checkpoint`), backed by the gfx950 kernels behind the C
ABI of include/cocr.h.  torch holds the parameters (state-dict compatible with reference
checkpoints) and owns device memory; no layer of the network is evaluated by torch.
"""
from __future__ import annotations

import io
import json
import logging
import tarfile
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .codec import PytorchCodec
from .ctc_decoder import BeamDecoder, GreedyDecoder, greedy_decoder
from .engine import HipRecognizer
from .spec import HParams, encoder_state_spec

logger = logging.getLogger(__name__)


class _Holder(nn.Module):
    """Parameter container: no forward, only the reference's module tree for its state-dict keys."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameters only: the network runs in libcocr_hip.so')


def _build_param_tree(spec) -> nn.Module:
    root = _Holder()
    for name, (shape, kind) in spec.items():
        parts = name.split('.')
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Holder())
            mod = getattr(mod, p)
        leaf = parts[-1]
        if kind == 'param':
            if len(shape) >= 2:
                t = torch.empty(shape).uniform_(-0.05, 0.05)
            else:
                t = torch.ones(shape) if leaf == 'weight' else torch.zeros(shape)
            mod.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        elif kind == 'buffer':
            mod.register_buffer(leaf, torch.ones(shape) if leaf == 'running_var' else torch.zeros(shape))
        else:
            mod.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
    return root


class PytorchRecognitionModel(nn.Module):
    def __init__(self,
                 num_classes: int,
                 height: int,
                 encoder_dim: int,
                 num_encoder_layers: int,
                 num_attention_heads: int,
                 feed_forward_expansion_factor: int,
                 conv_expansion_factor: int,
                 input_dropout_p: float,
                 feed_forward_dropout_p: float,
                 attention_dropout_p: float,
                 conv_dropout_p: float,
                 conv_kernel_size: int,
                 half_step_residual: bool,
                 subsampling_conv_channels: int,
                 subsampling_factor: int,
                 codec: PytorchCodec,
                 ctc_decoder=greedy_decoder,
                 **kwargs):
        """
        Inference version of a conformer_ocr RecognitionModel (pred.py:51-98).  Dropout
        probabilities are accepted and ignored (inference); unknown keyword arguments are ignored
        like in the reference, except `compute_dtype` ('bf16' default, or 'fp32') and `chain_rows`.
        """
        super().__init__()
        self.hparams_record = HParams(num_classes=num_classes, height=height, encoder_dim=encoder_dim,
                                      num_encoder_layers=num_encoder_layers, num_attention_heads=num_attention_heads,
                                      feed_forward_expansion_factor=feed_forward_expansion_factor,
                                      conv_expansion_factor=conv_expansion_factor, conv_kernel_size=conv_kernel_size,
                                      half_step_residual=bool(half_step_residual),
                                      subsampling_conv_channels=subsampling_conv_channels,
                                      subsampling_factor=subsampling_factor)
        encoder = _build_param_tree(encoder_state_spec(self.hparams_record))
        decoder = nn.Linear(encoder_dim, num_classes, bias=True)
        for p in decoder.parameters():
            p.requires_grad_(False)
        self.nn = nn.ModuleDict({'encoder': encoder,
                                 'decoder': decoder})

        self.codec = codec
        self.ctc_decoder = ctc_decoder
        self.height = height
        self.channels = 1
        self.width = 0
        self.compute_dtype = kwargs.get('compute_dtype', 'bf16')
        # the reference's four dropout probabilities: identity at inference, used by conformer_ocr_amd.train.Trainer
        self.dropout_p = (float(input_dropout_p), float(feed_forward_dropout_p), float(attention_dropout_p), float(conv_dropout_p))
        # rows per workgroup of the row-chain kernels (include/cocr.h cocr_set_chain_rows).  This class is called one batch at a time
        # (the reference's loop, cli/test.py:185-199): 48-row blocks give every CU a workgroup at 32 x 300 frames and the shortest
        # forward; a caller that keeps several batches in flight on its own streams (bench.py) passes chain_rows=0.
        self.chain_rows = int(kwargs.get('chain_rows', 48))
        self._engine: Optional[HipRecognizer] = None
        self._engine_sig = None
        self._sig_tensors = None

    # ------------------------------------------------------------------ device model
    def _apply(self, fn, recurse=True):
        """`.to()` / `.cuda()` / `.float()`: parameters get new storage, buffers are replaced -- the cached tensor list is rebuilt."""
        out = super()._apply(fn, recurse)
        self._sig_tensors = None
        return out

    def invalidate_engine(self) -> None:
        """Forces a re-pack of the device model at the next call.  Not needed after any of: in-place writes, `load_state_dict`
        (also with `assign=True`), `.to()`, `p.data = new`, replacing a parameter, a buffer, a sub-module
        (`net.nn.decoder = nn.Linear(...)`) or `net.nn` itself -- `_signature` sees all of these.  What it cannot see: a write
        that bypasses the version counter of a tensor whose address stays the same (raw pointer writes from another library,
        `tensor.data.copy_()` counts and is seen)."""
        self._sig_tensors = None
        self._engine_sig = None

    def _signature(self, device: torch.device):
        """Identity of the weights the packed device model was built from, re-read on EVERY call:
        * which objects hang in the module tree: ids of every module's children, parameters and buffers (a replaced sub-module,
          `load_state_dict(assign=True)`, `del` / re-register) -- read from the modules' own dicts, not through `state_dict()`
          or `parameters()` (1.5 ms / 0.6 ms per call for the 12-block model);
        * where their values live: `data_ptr()` of every tensor (`p.data = new`, `.to()`);
        * whether they were written in place: the sum of the version counters.
        About 0.15 ms for 478 tensors in 539 modules; `forward` runs it AFTER it has enqueued the launch (the GPU is busy for
        0.8 ms or more) and repeats the call in the rare case that the answer is "changed"."""
        ids = None
        if self._sig_tensors is not None and self._sig_tensors[3][0] is self.nn:      # (a replaced `self.nn`: the cached dicts are the old tree's)
            ids = [id(v) for d in self._sig_tensors[1] for v in d.values()]
        if ids is None or ids != self._sig_tensors[2]:
            # first call, or the tree changed.  The cache keeps the modules and tensors it describes alive, so an id in `ids`
            # cannot have been handed to a new object in the meantime; the generation counts the rebuilds.
            mods = list(self.nn.modules())
            dicts = [d for m in mods for d in (m._modules, m._parameters, m._buffers)]
            self._sig_tensors = ([*self.nn.parameters(), *self.nn.buffers()], dicts, [id(v) for d in dicts for v in d.values()], mods)
            self._sig_gen = getattr(self, '_sig_gen', 0) + 1
        ts = self._sig_tensors[0]
        ver, ptrs = 0, []
        for t in ts:
            ver += t._version
            ptrs.append(t.data_ptr())
        return (str(device), self.compute_dtype, self._sig_gen, id(self.nn), tuple(ptrs), ver)

    def _unchanged_at_a_glance(self) -> bool:
        """The cheap part of `_signature` (about 40 us): the same module tree object and the same sum of version counters as when
        the device model was packed.  In-place writes (every optimiser step of a training / validation interleave) show here;
        `forward` then re-packs BEFORE launching instead of running a whole forward on stale weights first."""
        st, sig = self._sig_tensors, self._engine_sig
        if st is None or sig is None or st[3][0] is not self.nn:
            return False
        ver = 0
        for t in st[0]:
            ver += t._version
        return ver == sig[-1]

    def _pack(self, device: torch.device, sig) -> HipRecognizer:
        eng = HipRecognizer(self.hparams_record, device, self.compute_dtype)
        eng.load_state({k: v for k, v in self.nn.state_dict().items()}, strict=True)
        eng.finalize()
        eng.set_chain_rows(self.chain_rows)
        eng.set_graph(True)        # hipGraph replay of the launch sequence, staged for the fresh tensors every call brings
        self._engine, self._engine_sig = eng, sig
        return eng

    def engine(self, device: Optional[torch.device] = None) -> HipRecognizer:
        """The packed device model, (re)built when the parameters or the device changed."""
        if device is None:
            device = next(self.nn.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('PytorchRecognitionModel (HIP) needs a GPU: move the model / line batch to a cuda device; '
                               'there is no CPU fallback')
        sig = self._signature(device)
        if self._engine is None or sig != self._engine_sig:
            self._pack(device, sig)
        return self._engine

    def engine_pool(self, n: int, device: Optional[torch.device] = None) -> List[HipRecognizer]:
        """`n` packed copies of the device model for callers that keep several batches in flight, one per stream
        (conformer_ocr_amd/evaluate.py `recognize(..., streams=n)`): each has its own workspace and captured launch sequences and reads the
        first engine's weights (`cocr_share_weights`); the
        row-chain kernels run their throughput form (96-row blocks: every weight byte streamed once per block -- with several batches
        in flight the chip is full anyway).  Rebuilt when the parameters change, like `engine()`."""
        from . import hw_queues_note
        hw_queues_note(n)
        eng0 = self.engine(device)
        if getattr(self, '_pool_sig', None) != self._engine_sig:
            self._pool, self._pool_sig = [], self._engine_sig
        while len(self._pool) < n:
            eng = HipRecognizer(self.hparams_record, eng0.device, self.compute_dtype)
            eng.share_weights(eng0)        # one set of weights and tables for all copies (they stay in the Infinity Cache; include/cocr.h)
            eng.set_chain_rows(0)
            eng.set_graph(True)
            self._pool.append(eng)
        return self._pool[:n]

    def pool_streams(self, n: int, device: Optional[torch.device] = None) -> List['torch.cuda.Stream']:
        """The streams the copies of `engine_pool` run on: kept, so that repeated loops queue on the same few hardware queues."""
        dev = self.engine(device).device
        if getattr(self, '_pool_streams_dev', None) != dev:
            self._pool_streams_list, self._pool_streams_dev = [], dev
        while len(self._pool_streams_list) < n:
            self._pool_streams_list.append(torch.cuda.Stream(dev))
        return self._pool_streams_list[:n]

    def adopt_engine(self, eng: HipRecognizer) -> None:
        """Use an already packed device model (e.g. one whose weights arrived by RCCL broadcast)."""
        self._engine = eng
        self._engine_sig = self._signature(eng.device)

    # ------------------------------------------------------------------ reference surface
    def forward(self, line: torch.Tensor, lens: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        Performs a forward pass on a torch tensor of one or more lines with
        shape (N, C, H, W) and returns the logits (N, W', num_classes) [a torch tensor on the
        line's device, as the reference returns despite its docstring] and the int32 output
        sequence lengths (pred.py:101-122).  `lens` is required, as in the reference
        (calc_length(None) raises there).
        """
        if lens is None:
            raise TypeError('lens is required (the reference raises in calc_length for lens=None)')
        if line.dim() != 4 or line.shape[1] != 1:
            raise ValueError(f'expected a (N,1,H,W) line batch, got {tuple(line.shape)}')
        lens_np = torch.as_tensor(lens).cpu().numpy()
        eng = self._engine
        if eng is None or not line.is_cuda or line.device != eng.device:
            eng = self.engine(line.device if line.is_cuda else None)
            probits, out_lens = eng.forward(line.squeeze(1), lens_np)
        elif not self._unchanged_at_a_glance():
            # version counters moved (or the tree object changed): pack first, then launch -- no forward on stale weights
            eng = self.engine(eng.device)
            probits, out_lens = eng.forward(line.squeeze(1), lens_np)
        else:
            # launch first, walk the whole tree while the GPU works (`_signature`: ids, addresses, versions of 478 tensors -- in front
            # of every launch that was 10 % of a 32-line call); the cheap glance above has already caught in-place writes
            with torch.no_grad():
                probits, out_lens = eng.forward(line.squeeze(1), lens_np)
            sig = self._signature(eng.device)
            if sig != self._engine_sig:                    # a replaced tensor or sub-module: pack again, run again
                torch.cuda.current_stream(eng.device).synchronize()      # (the old engine's launches are still queued on its buffers)
                eng = self._pack(eng.device, sig)
                probits, out_lens = eng.forward(line.squeeze(1), lens_np)
        return probits, torch.from_numpy(out_lens)

    def transform_lines(self, crops, pad: int = 16, bucket_edge: int = 0, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The step kraken's `ImageInputTransforms(1, height, 0, 1, (pad, 0))` performs in the reference's data pipeline
        (dataset.py:89, cli/test.py:156), on the GPU: raw 8-bit line crops ((H, W) or (H, W, 3) numpy arrays of any height) ->
        a `(N, 1, height, W)` uint8 line batch (pixel / 255 = the [0, 1] floats `forward` expects; it ingests uint8 directly)
        and the `seq_lens` (scaled width + 2 pad).  W = the widest line, rounded up to a multiple of `bucket_edge` if given."""
        eng = self.engine(device)
        batch, lens = eng.preprocess(list(crops), height=self.height, pad=pad, bucket_edge=bucket_edge)
        return batch.unsqueeze(1), torch.from_numpy(lens)

    def _label_records(self, line: torch.Tensor, lens: torch.Tensor) -> List[List[Tuple[int, int, int, float]]]:
        o, olens = self.forward(line, lens)
        if isinstance(self.ctc_decoder, GreedyDecoder):
            return self._engine.ctc_greedy(o, olens.numpy())
        if isinstance(self.ctc_decoder, BeamDecoder):
            return self._engine.ctc_beam(o, olens.numpy(), self.ctc_decoder.beam_size)
        # a user-supplied decoder: the reference's own host loop (pred.py:158-162)
        o = o.transpose(1, 2).cpu().float().numpy()
        return [self.ctc_decoder(seq[:, :int(seq_len)]) for seq, seq_len in zip(o, olens)]

    def predict(self, line: torch.Tensor, lens: Optional[torch.Tensor] = None) -> List[List[Tuple[str, int, int, float]]]:
        """
        Forward pass + decoding as lists of (string, start, end, confidence) tuples, one list per
        line -- what the reference's docstring promises (pred.py:124-136; its body calls the codec
        object itself, which kraken's codec does not support).
        """
        return [self.codec.decode(locs) for locs in self._label_records(line, lens)]

    def predict_string(self, line: torch.Tensor, lens: Optional[torch.Tensor] = None) -> List[str]:
        """
        Forward pass on a (N, C, H, W) batch; returns one string per line, batch order
        preserved (pred.py:148-164).
        """
        lut = self._codec_lut()
        if lut is not None and isinstance(self.ctc_decoder, (GreedyDecoder, BeamDecoder)):
            # device decode + a 1:1 codec: label arrays -> characters by one table lookup per line (the record lists of `predict_labels`
            # and the codec's tuple walk cost 0.26 ms per 32 lines on the host, after the GPU had finished: a fifth of the call)
            o, olens = self.forward(line, lens)
            eng = self._engine
            if isinstance(self.ctc_decoder, GreedyDecoder):
                h = eng.ctc_greedy_async(o, olens.numpy())
            else:
                h = eng._decode_async(eng.lib.cocr_ctc_beam, o, olens.numpy(), extra=(int(self.ctc_decoder.beam_size),))
            size = lut.shape[0]
            return [''.join(lut[np.minimum(lab, size - 1)].tolist()) for lab in eng.collect_labels(h)]
        return [''.join(x[0] for x in self.codec.decode(locs)) for locs in self._label_records(line, lens)]

    def _codec_lut(self):
        """label -> character table of a 1:1, non-strict codec ('' for labels the codec does not know: skipped, as `decode` skips them;
        last entry: every label beyond the table); None for codecs with multi-label graphemes."""
        c = self.codec
        l2c1 = getattr(c, '_l2c1', None)
        if l2c1 is None or getattr(c, 'strict', False):
            return None
        key = hash(tuple(l2c1.items()))                 # (an in-place edit of the codec's map that keeps its size must rebuild the table)
        if getattr(self, '_lut_of', None) is not c or getattr(self, '_lut_key', None) != key:
            top = max(l2c1, default=0)
            if top > (1 << 20):                         # sparse label ids: no table, `decode()` does it
                self._lut = None
            else:
                width = max((len(v) for v in l2c1.values()), default=1)
                lut = np.zeros(top + 2, dtype=f'<U{max(width, 1)}')
                for k, v in l2c1.items():
                    lut[k] = v
                self._lut = lut
            self._lut_of, self._lut_key = c, key
        return self._lut

    def predict_string_async(self, line: torch.Tensor, lens: torch.Tensor):
        """`predict_string` split in two for pipelined callers (conformer_ocr_amd/evaluate.py): enqueues forward, CTC decode and
        the read-back of the label records on the current stream and returns a handle; `collect_strings(handle)` waits for
        that batch only -- the next batch's upload and forward can be in flight meanwhile."""
        o, olens = self.forward(line, lens)
        eng = self._engine             # the handle carries the engine that owns its pinned records (a re-pack may replace `self._engine` before the collect)
        if isinstance(self.ctc_decoder, GreedyDecoder):
            return ('device', eng.ctc_greedy_async(o, olens.numpy()), eng)
        if isinstance(self.ctc_decoder, BeamDecoder):
            return ('device', eng._decode_async(eng.lib.cocr_ctc_beam, o, olens.numpy(), extra=(int(self.ctc_decoder.beam_size),)), eng)
        o = o.transpose(1, 2).cpu().float().numpy()           # a user-supplied decoder: the reference's own host loop
        return ('host', [self.ctc_decoder(seq[:, :int(seq_len)]) for seq, seq_len in zip(o, olens)], None)

    def collect_strings(self, handle) -> List[str]:
        kind, h, eng = handle
        lut = self._codec_lut() if kind == 'device' else None
        if lut is not None:
            size = lut.shape[0]
            return [''.join(lut[np.minimum(lab, size - 1)].tolist()) for lab in eng.collect_labels(h)]
        records = eng.collect(h) if kind == 'device' else h
        return [''.join(x[0] for x in self.codec.decode(locs)) for locs in records]

    def step(self, batch: Dict, with_grad: bool = False) -> Dict:
        """The reference's `RecognitionModel._step` (model.py:129-145) on a batch dict {'image' (N,1,H,W), 'seq_lens' (N),
        'target' (sum target_lens,) concatenated labels, 'target_lens' (N)}: forward, then the criterion
        `nn.CTCLoss(reduction='sum', zero_infinity=True)` on `log_softmax(probits)` -- on the device, without materialising the
        log-probabilities.  Returns {'loss' (0-dim device tensor), 'probits', 'output_lens'} like the reference, plus 'nll'
        (per-line terms of the sum) and, with_grad, 'grad_probits' = d loss / d probits: what autograd hands to the decoder's
        backward in `training_step` (model.py:147-152).  Eval-mode forward (BatchNorm running statistics, no dropout): this is the
        reference's `validation_step` forward (model.py:154-156); the encoder's backward is not part of this library yet."""
        probits, out_lens = self.forward(batch['image'], batch['seq_lens'])
        nll, grad = self._engine.ctc_loss(probits, out_lens.numpy(), torch.as_tensor(batch['target']).cpu().numpy(),
                                          torch.as_tensor(batch['target_lens']).cpu().numpy(), with_grad=with_grad)
        out = {'loss': nll.sum(), 'probits': probits, 'output_lens': out_lens, 'nll': nll}
        if with_grad:
            out['grad_probits'] = grad
        return out

    def training_step(self, batch: Dict, batch_idx: int = 0) -> torch.Tensor:
        """The reference's `RecognitionModel.training_step` (model.py:147-152): train-mode forward (BatchNorm batch statistics, dropout
        with the constructor's probabilities), summed CTC loss -- returned as a DIFFERENTIABLE 0-dim device tensor: `loss.backward()`
        leaves `.grad` on every parameter of `self.nn` that requires grad, and any torch optimizer / Lightning loop drives the step
        (model.py:283-289).  Forward and backward run in libcocr_hip.so (conformer_ocr_amd/autograd.py); fp32, or bf16-rounded
        matmul operands with `self.matmul_precision = 'medium'` (the reference's cli/train.py:252)."""
        if not any(p.requires_grad for p in self.nn.parameters()):
            raise RuntimeError('no parameter of net.nn requires grad: call net.nn.requires_grad_(True) (the holders are created frozen for inference)')
        from .autograd import training_step
        return training_step(self, batch)

    def predict_labels(self, line: torch.tensor, lens: torch.Tensor = None) -> List[List[Tuple[int, int, int, float]]]:
        """
        Forward pass on a (N, C, H, W) batch; returns per line a list of tuples
        (class, start, end, max) (pred.py:166-178).
        """
        return self._label_records(line, lens)

    @classmethod
    def load_safetensors(cls, path, **kwargs):
        """
        Loads a safetensors archive: tar with `metadata.json` {codec, hyper_params} and
        `model.safetensors` = `nn.state_dict()` (pred.py:180-195).
        """
        import safetensors.torch
        with tarfile.open(path, 'r') as tf:
            metadata = json.load(tf.extractfile('metadata.json'))
            if 'codec' not in metadata:
                raise ValueError('No codec in metadata record')
            codec = PytorchCodec(metadata['codec'])
            if 'hyper_params' not in metadata:
                raise ValueError('No hyperparameters in metadata record')
            net = cls(**{**metadata['hyper_params'], **kwargs}, codec=codec)
            weights = safetensors.torch.load(tf.extractfile('model.safetensors').read())
        net.nn.load_state_dict(weights)
        return net.eval()

    @classmethod
    def load_checkpoint(cls, path, **kwargs):
        """
        Loads a lightning checkpoint (pred.py:197-210): `state_dict` (keys `nn.encoder.*`,
        `nn.decoder.*`), `hyper_parameters`, `TextLineDataModule.codec`.
        """
        state_dict = torch.load(path, map_location='cpu', weights_only=True)
        if 'TextLineDataModule' not in state_dict:
            raise ValueError('Checkpoint does not contain data module state.')
        codec = PytorchCodec(state_dict['TextLineDataModule']['codec'])
        if 'hyper_parameters' not in state_dict:
            raise ValueError('No hyperparameters in state_dict')
        net = cls(**{**state_dict['hyper_parameters'], **kwargs}, codec=codec)
        net.load_state_dict(state_dict['state_dict'], strict=False)
        return net.eval()


def save_safetensors(net: PytorchRecognitionModel, path, extra_metadata: Optional[Dict] = None) -> None:
    """Writes the archive format `load_safetensors` reads (the reference's converter, pred.py:213-254,
    does not run: SURVEY A.4)."""
    import safetensors.torch
    hp = net.hparams_record.as_dict()
    hp.update(input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1)
    meta = {'codec': net.codec.c2l, 'hyper_params': hp}
    meta.update(extra_metadata or {})
    blob = safetensors.torch.save({k: v.detach().cpu().contiguous() for k, v in net.nn.state_dict().items()})
    mj = json.dumps(meta).encode()
    with tarfile.open(path, 'w') as tf:
        for name, data in (('model.safetensors', blob), ('metadata.json', mj)):
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))


def average_checkpoints(paths, num_checkpoints: Optional[int] = None) -> Dict:
    """The reference's `avg_ckpts` (cli/train.py:37-98): element-wise mean of the `state_dict`s of the last `num_checkpoints`
    checkpoints of `paths` (sorted by name like the reference's glob); everything else is taken from the first of them.
    Floating tensors: sum, then in-place division; integer tensors (BatchNorm `num_batches_tracked`): floor division;
    half tensors are summed in float32; a checkpoint whose key list differs raises KeyError.  Returns the averaged checkpoint
    dict (`torch.save` it, or hand `['state_dict']` to `load_state_dict`)."""
    import collections
    ckpts = sorted(str(p) for p in paths)
    if num_checkpoints is not None:
        if num_checkpoints < 2:
            raise ValueError('at least 2 checkpoints are averaged')
        if len(ckpts) < num_checkpoints:
            raise ValueError(f'Less checkpoints found than requested for averaging ({len(ckpts)} < {num_checkpoints})')
        ckpts = ckpts[-num_checkpoints:]
    params, keys, new_state = collections.OrderedDict(), None, None
    for fpath in ckpts:
        state = torch.load(fpath, map_location='cpu', weights_only=True)
        if new_state is None:
            new_state = state
        sd = state['state_dict']
        if keys is None:
            keys = list(sd.keys())
        elif keys != list(sd.keys()):
            raise KeyError(f'For checkpoint {fpath}, expected list of params: {keys}, but found: {list(sd.keys())}')
        for k in keys:
            p = sd[k]
            if p.dtype == torch.float16:
                p = p.float()
            if k not in params:
                params[k] = p.clone()
            else:
                params[k] += p
    for k, v in params.items():
        if v.is_floating_point():
            v.div_(len(ckpts))
        else:
            v //= len(ckpts)
    new_state['state_dict'] = params
    return new_state
