"""
Hyper-parameter record and state-dict key map of the recognition path.

The names are the reference's own: the constructor arguments of
``PytorchRecognitionModel`` (reference conformer_ocr/pred.py:51-69) and the
state-dict keys its ``nn`` ModuleDict produces (encoder.py:67-103,
convolution.py:192-225, attention.py:59-70, feed_forward.py:45-52,
pred.py:89-91).  Nothing here computes; it only says which tensors exist and
what shape they have, so that the weight loader, the synthetic weight
generator and the C-ABI upload all agree with a reference checkpoint.
"""
from __future__ import annotations

import dataclasses
import math
from collections import OrderedDict
from typing import Dict, Tuple

# constructor arguments that shape the network (pred.py:51-66); the dropout
# probabilities are accepted and ignored (inference: identity).
HPARAM_NAMES = ('num_classes', 'height', 'encoder_dim', 'num_encoder_layers',
                'num_attention_heads', 'feed_forward_expansion_factor',
                'conv_expansion_factor', 'conv_kernel_size', 'half_step_residual',
                'subsampling_conv_channels', 'subsampling_factor')

POS_TABLE_MAX_LEN = 5000   # RelPositionalEncoding(max_len=5000), embedding.py:29
LN_EPS = 1e-5              # torch default, never overridden by the reference
BN_EPS = 1e-5


@dataclasses.dataclass(frozen=True)
class HParams:
    num_classes: int
    height: int = 96
    encoder_dim: int = 144
    num_encoder_layers: int = 16
    num_attention_heads: int = 4
    feed_forward_expansion_factor: int = 4
    conv_expansion_factor: int = 2
    conv_kernel_size: int = 31
    half_step_residual: bool = True
    subsampling_conv_channels: int = 32
    subsampling_factor: int = 4

    def __post_init__(self):
        # the same checks the reference modules make at construction time
        if self.encoder_dim % self.num_attention_heads:
            raise AssertionError('d_model % num_heads should be zero.')          # attention.py:53
        if (self.conv_kernel_size - 1) % 2:
            raise AssertionError("kernel_size should be a odd number for 'SAME' padding")  # convolution.py:132
        if self.conv_expansion_factor != 2:
            raise AssertionError('Currently, Only Supports expansion_factor 2')  # convolution.py:133
        if not math.log(self.subsampling_factor, 2).is_integer():
            raise ValueError('Sampling factor should be a power of 2.')          # convolution.py:177-178

    # -- derived sizes ------------------------------------------------------
    @property
    def sampling_num(self) -> int:
        return int(math.log(self.subsampling_factor, 2))

    @property
    def d_head(self) -> int:
        return self.encoder_dim // self.num_attention_heads

    @property
    def out_feats(self) -> int:
        """Height after the stride-2 stages (calc_length on `height`, convolution.py:217-222)."""
        return out_len(self.height, self.sampling_num)

    @property
    def ff_residual_factor(self) -> float:
        return 0.5 if self.half_step_residual else 1.0                          # encoder.py:62-65

    @classmethod
    def from_kwargs(cls, **kw) -> 'HParams':
        """Build from a checkpoint's `hyper_parameters` dict; unknown keys are ignored (pred.py:69)."""
        return cls(**{k: kw[k] for k in HPARAM_NAMES if k in kw})

    def as_dict(self) -> Dict:
        return dataclasses.asdict(self)


def out_len(length: int, repeat: int = 2) -> int:
    """Integer form of calc_length (convolution.py:240-247) for k=3, s=2, p=1:
    floor((l + 2 - 3)/2 + 1) per stage == (l - 1)//2 + 1 for l >= 1."""
    for _ in range(repeat):
        length = (int(length) - 1) // 2 + 1 if length >= 1 else int(math.floor((length - 1) / 2 + 1))
    return int(length)


def encoder_state_spec(hp: HParams) -> 'OrderedDict[str, Tuple[Tuple[int, ...], str]]':
    """name -> (shape, kind) for every entry of `ConformerEncoder.state_dict()` in
    the reference's order.  kind is 'param', 'buffer' or 'counter' (int64)."""
    D, C, k, h = hp.encoder_dim, hp.subsampling_conv_channels, hp.conv_kernel_size, hp.num_attention_heads
    ff = hp.feed_forward_expansion_factor * D
    sd: 'OrderedDict[str, Tuple[Tuple[int, ...], str]]' = OrderedDict()

    def p(name, *shape):
        sd[name] = (tuple(shape), 'param')

    # Conv2dSubsampling (convolution.py:190-225): conv.0, ReLU, then per extra
    # stage (depthwise, pointwise, ReLU) -> indices 2,3 | 5,6 | ...
    p('conv_subsample.conv.0.weight', C, 1, 3, 3)
    p('conv_subsample.conv.0.bias', C)
    idx = 2
    for _ in range(hp.sampling_num - 1):
        p(f'conv_subsample.conv.{idx}.weight', C, 1, 3, 3)
        p(f'conv_subsample.conv.{idx}.bias', C)
        p(f'conv_subsample.conv.{idx + 1}.weight', C, C, 1, 1)
        p(f'conv_subsample.conv.{idx + 1}.bias', C)
        idx += 3
    p('conv_subsample.out.0.weight', D, C * hp.out_feats)
    p('conv_subsample.out.0.bias', D)
    for l in range(hp.num_encoder_layers):
        pre = f'layers.{l}.sequential.'
        for ffn in (0, 3):                                                       # encoder.py:68-75, 91-98
            q = f'{pre}{ffn}.module.sequential.'
            p(q + '0.weight', D)
            p(q + '0.bias', D)
            p(q + '1.linear.weight', ff, D)
            p(q + '1.linear.bias', ff)
            p(q + '4.linear.weight', D, ff)
            p(q + '4.linear.bias', D)
            if ffn == 0:
                a = f'{pre}1.module.'                                            # encoder.py:76-82
                p(a + 'layer_norm.weight', D)
                p(a + 'layer_norm.bias', D)
                p(a + 'attention.u_bias', h, hp.d_head)
                p(a + 'attention.v_bias', h, hp.d_head)
                for proj in ('query', 'key', 'value'):
                    p(a + f'attention.{proj}_proj.linear.weight', D, D)
                    p(a + f'attention.{proj}_proj.linear.bias', D)
                p(a + 'attention.pos_proj.linear.weight', D, D)                  # bias=False, attention.py:62
                p(a + 'attention.out_proj.linear.weight', D, D)
                p(a + 'attention.out_proj.linear.bias', D)
                c = f'{pre}2.module.sequential.'                                 # encoder.py:83-90
                p(c + '0.weight', D)
                p(c + '0.bias', D)
                p(c + '2.conv.weight', 2 * D, D, 1)
                p(c + '2.conv.bias', 2 * D)
                p(c + '4.conv.weight', D, 1, k)                                  # depthwise, bias=False
                p(c + '5.weight', D)
                p(c + '5.bias', D)
                sd[c + '5.running_mean'] = ((D,), 'buffer')
                sd[c + '5.running_var'] = ((D,), 'buffer')
                sd[c + '5.num_batches_tracked'] = ((), 'counter')
                p(c + '7.conv.weight', D, D, 1)
                p(c + '7.conv.bias', D)
        p(f'{pre}4.weight', D)                                                   # encoder.py:99
        p(f'{pre}4.bias', D)
    return sd


def model_state_spec(hp: HParams) -> 'OrderedDict[str, Tuple[Tuple[int, ...], str]]':
    """Keys of `PytorchRecognitionModel.nn.state_dict()` (pred.py:89-91): `encoder.*` + `decoder.*`."""
    sd: 'OrderedDict[str, Tuple[Tuple[int, ...], str]]' = OrderedDict()
    for n, v in encoder_state_spec(hp).items():
        sd['encoder.' + n] = v
    sd['decoder.weight'] = ((hp.num_classes, hp.encoder_dim), 'param')
    sd['decoder.bias'] = ((hp.num_classes,), 'param')
    return sd


def flops_per_line(hp: HParams, width: int) -> float:
    """Algorithmic FLOP (2 x MAC) of one line of `width` pixels through the path
    (BASELINE.md section 3 formula; positional projection excluded)."""
    assert hp.sampling_num == 2, 'formula is stated for subsampling_factor 4'
    C, D, L, k = hp.subsampling_conv_channels, hp.encoder_dim, hp.num_encoder_layers, hp.conv_kernel_size
    F1 = out_len(hp.height, 1)
    F = hp.out_feats
    T1 = out_len(width, 1)
    T = out_len(width, 2)
    ffe = hp.feed_forward_expansion_factor
    mac = 9 * C * T1 * F1 + 9 * C * T * F + C * C * T * F + C * F * D * T
    mac += L * (2 * (2 * ffe * D * D) * T + 4 * D * D * T + 3 * T * T * D + 2 * D * D * T + k * D * T + D * D * T)
    mac += D * hp.num_classes * T
    return 2.0 * mac
