"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's
recognition forward (line batch -> conformer encoder -> linear decoder).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product path (conformer_ocr_amd/) never
does and fails loudly without its HIP library.

Written from the closed forms of SURVEY.md Appendix A, not from the reference's
module code: native (B,H,W) input, channel-last activations, BatchNorm folded
into the depthwise taps, the positional projection taken from the full
9999-row table, integer length arithmetic.  Each function cites the reference
file:line (relative to /root/reference/conformer_ocr) whose result it restates.

Pinning: `tests/golden/make_golden.py` imports the reference encoder itself in
the authoring container and stores its outputs (final and per stage) for the
tiny / cfg1 configurations; `tests/test_oracle.py` checks this restatement
against those fixtures (fp32: <= 2e-5 abs on logits).  The CTC decoders live in
`ctc_ref.py` (third-party kraken algorithm: parity unpinned, see there).

Everything takes and returns torch CPU tensors; `dtype` selects float32 (the
reference's inference precision, cli/test.py:107) or float64.

`bf16_operands=True` is the checker for the HIP library's bf16 compute mode: the
same arithmetic with every matrix-product operand -- weights and activations --
rounded to bfloat16 at the points where the kernels round them (DESIGN.md
section 2: the residual stream, LayerNorm statistics, biases, softmax and all
accumulation stay fp32; pixels, conv taps, Z1/Z2/Z3, the normalised operand xn,
the FFN hidden, q/k/v, (q+u)/(q+v) pre-scaled, the positional table, the
softmax numerators, ctx, GLU and depthwise outputs are bf16).  What is left
between this mode and the kernels is accumulation order and the hardware's
exp2/rcp approximations, so the comparison tolerance is an order of magnitude
below the bf16-vs-fp32 band.  It restates the library's documented rounding
points, not the reference: it is pinned only through the fp32 mode it shares
all indexing with.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5
POS_MAX_LEN = 5000


def out_len(length, repeat: int = 2):
    """calc_length, conformer/convolution.py:240-247 with k=3, s=2, p=1 (float math of the
    reference reduces to (l-1)//2 + 1 per stage for integer l >= 1)."""
    l = torch.as_tensor(length).to(torch.int64)
    for _ in range(repeat):
        l = torch.div(l - 1, 2, rounding_mode='floor') + 1
    return l.to(torch.int32)


def sinusoid_table(d_model: int, max_len: int = POS_MAX_LEN) -> torch.Tensor:
    """conformer/embedding.py:35-56: rows 0..2*max_len-2 hold PE(p) for p = max_len-1 ... -(max_len-1);
    PE(p)[2m] = sin(p w_m), PE(p)[2m+1] = cos(p w_m), w_m = exp(-2m ln(1e4)/d).  float32 like the reference."""
    pos = torch.arange(max_len - 1, -max_len, -1, dtype=torch.float32).unsqueeze(1)      # +L-1 ... -(L-1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * max_len - 1, d_model, dtype=torch.float32)
    # the reference evaluates sin(-1 * position * div_term) for the negative half (embedding.py:49-50);
    # (-p) * w == -(p * w) exactly in floating point, so one signed product restates both halves
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class Oracle:
    """Functional forward over a plain state dict (keys `encoder.*`, `decoder.*`)."""

    def __init__(self, hp, state: Mapping[str, np.ndarray], dtype=torch.float32, bf16_operands: bool = False,
                 fused_frontend: bool = True):
        self.hp = hp
        self.dtype = dtype
        self.training = False          # train mode (`forward_train`): BatchNorm batch statistics; dropout probabilities 0
        self.bn_batch_stats: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.bf16 = bool(bf16_operands)
        # bf16 mode: the fused frontend kernel (256 conv channels) also rounds pixels, the 3x3 taps and Z1; the
        # stand-alone conv kernel of the other channel counts keeps those fp32 and rounds Z2 only
        self.fused_frontend = bool(fused_frontend)
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in state.items():
            t = torch.as_tensor(np.asarray(v))
            t = t.to(dtype) if t.is_floating_point() else t
            if self.bf16 and t.is_floating_point() and self._is_product_weight(k, t):
                t = self.r(t)
            self.w[k] = t
        self._maxlen = POS_MAX_LEN
        self._pe = sinusoid_table(hp.encoder_dim).to(dtype)
        self._ptab: Dict[int, torch.Tensor] = {}

    def _is_product_weight(self, key: str, t: torch.Tensor) -> bool:
        """Weights that are MFMA operands in the library's bf16 mode: every Linear / pointwise-conv matrix, and the
        frontend's 3x3 taps on the fused path.  The conv module's depthwise taps (folded with BatchNorm), biases,
        LayerNorm / BatchNorm parameters, u_bias / v_bias stay fp32; pos_proj is applied in fp32 and its RESULT rounded."""
        if t.dim() < 2 or key.endswith('u_bias') or key.endswith('v_bias') or 'pos_proj' in key:
            return False
        if '.sequential.2.module.sequential.4.conv.weight' in key:       # conv-module depthwise taps
            return False
        if 'conv_subsample.conv.' in key and t.dim() == 4 and t.shape[1] == 1:      # 3x3 taps (conv.0 and depthwise stages)
            return self.fused_frontend
        return True

    def r(self, x: torch.Tensor) -> torch.Tensor:
        """Round to bfloat16 (nearest even) and back -- identity outside the bf16-operand mode."""
        return x.to(torch.bfloat16).to(self.dtype) if self.bf16 else x

    # ------------------------------------------------------------------ frontend
    def front_conv12(self, x_bhw: torch.Tensor) -> torch.Tensor:
        """F1+F2 (convolution.py:192-205 applied to the (B,1,W,H) view of pred.py:119 / convolution.py:233).
        Z1[b,c,t,f] = relu(b0[c] + sum w0[c,0,dt,df] X[b, 2f+df-1, 2t+dt-1]);
        Z2[b,c,t,f] = b2[c] + sum w2[c,0,dt,df] Z1[b,c,2t+dt-1,2f+df-1].   Returns Z2 as (B,T,F,C)."""
        w = self.w
        rf = self.r if self.fused_frontend else (lambda t: t)
        x = rf(x_bhw.to(self.dtype)).transpose(1, 2).unsqueeze(1)             # (B,1,W,H): conv rows run along image width
        z1 = rf(F.relu(F.conv2d(x, w['encoder.conv_subsample.conv.0.weight'], w['encoder.conv_subsample.conv.0.bias'],
                                stride=2, padding=1)))
        C = z1.shape[1]
        z2 = F.conv2d(z1, w['encoder.conv_subsample.conv.2.weight'], w['encoder.conv_subsample.conv.2.bias'],
                      stride=2, padding=1, groups=C)
        return self.r(z2.permute(0, 2, 3, 1).contiguous())                    # (B,T,F,C)

    def front_pw(self, z2_btfc: torch.Tensor, idx: int = 3) -> torch.Tensor:
        """F3 (convolution.py:207-213): Z3[b,t,f,o] = relu(b3[o] + sum_c w3[o,c] Z2[b,t,f,c])."""
        w3 = self.w[f'encoder.conv_subsample.conv.{idx}.weight'].flatten(1)   # (C,C)
        return self.r(F.relu(z2_btfc @ w3.t() + self.w[f'encoder.conv_subsample.conv.{idx}.bias']))

    def front_dw(self, z_btfc: torch.Tensor, idx: int) -> torch.Tensor:
        """extra stride-2 depthwise stage for subsampling_factor > 4 (convolution.py:200-205, loop body)."""
        z = z_btfc.permute(0, 3, 1, 2)
        C = z.shape[1]
        z = F.conv2d(z, self.w[f'encoder.conv_subsample.conv.{idx}.weight'],
                     self.w[f'encoder.conv_subsample.conv.{idx}.bias'], stride=2, padding=1, groups=C)
        return self.r(z.permute(0, 2, 3, 1).contiguous())

    def front_out(self, z3_btfc: torch.Tensor) -> torch.Tensor:
        """F4+F5 (convolution.py:224,235-236): Y[b,t,:] = Wout vec_{c,f}(Z3[b,:,t,:]) + bout, vec index c*F+f."""
        B, T, Fh, C = z3_btfc.shape
        v = z3_btfc.permute(0, 1, 3, 2).reshape(B, T, C * Fh)
        return v @ self.w['encoder.conv_subsample.out.0.weight'].t() + self.w['encoder.conv_subsample.out.0.bias']

    def frontend(self, x_bhw: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        if self.hp.sampling_num == 1:
            # factor 2: conv.0 + ReLU only
            w = self.w
            x = x_bhw.to(self.dtype).transpose(1, 2).unsqueeze(1)
            z = self.r(F.relu(F.conv2d(x, w['encoder.conv_subsample.conv.0.weight'], w['encoder.conv_subsample.conv.0.bias'],
                                       stride=2, padding=1)).permute(0, 2, 3, 1).contiguous())
        else:
            z2 = self.front_conv12(x_bhw)
            z = self.front_pw(z2, 3)
            if taps is not None:
                taps['front.z2'] = z2
                taps['front.z3'] = z
            idx = 5
            for _ in range(self.hp.sampling_num - 2):
                z = self.front_pw(self.front_dw(z, idx), idx + 1)
                idx += 3
        y = self.front_out(z)
        if taps is not None:
            taps['front.y'] = y
        return y

    # ------------------------------------------------------------------ block parts
    def _ln(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + '.weight'], self.w[prefix + '.bias'], LN_EPS)

    def _ln_op(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        """A LayerNorm whose result is the next product's operand (bf16 mode: the library's `xn` buffer)."""
        return self.r(self._ln(x, prefix))

    def ffn(self, y: torch.Tensor, l: int, which: int) -> torch.Tensor:
        """feed_forward.py:45-52 inside ResidualConnectionModule (modules.py:32, encoder.py:62-75):
        y + f * (W2 silu(W1 LN(y) + b1) + b2), f = 0.5 with half_step_residual."""
        p = f'encoder.layers.{l}.sequential.{which}.module.sequential.'
        t = self._ln_op(y, p + '0')
        t = self.r(F.silu(t @ self.w[p + '1.linear.weight'].t() + self.w[p + '1.linear.bias']))
        t = t @ self.w[p + '4.linear.weight'].t() + self.w[p + '4.linear.bias']
        return y + self.hp.ff_residual_factor * t

    def pos_table(self, l: int) -> torch.Tensor:
        """P_full = PE_full Wpos^T (attention.py:62,85 on embedding.py:66's table), 9999 x D; row 4999 <-> p = 0."""
        if l not in self._ptab:      # input-independent: computed once per layer, like a load-time constant
            p = f'encoder.layers.{l}.sequential.1.module.attention.'
            self._ptab[l] = self.r(self._pe @ self.w[p + 'pos_proj.linear.weight'].t())
        return self._ptab[l]

    def mhsa(self, y: torch.Tensor, l: int, taps: Optional[dict] = None) -> torch.Tensor:
        """attention.py:72-103,143-151:
        score[b,h,i,j] = ((q_i+u_h).k_j + (q_i+v_h).P_h[(T-1)-(i-j)]) / sqrt(dh), softmax over ALL j (no mask)."""
        hp = self.hp
        B, T, D = y.shape
        h, dh = hp.num_attention_heads, hp.d_head
        m = f'encoder.layers.{l}.sequential.1.module.'
        a = m + 'attention.'
        xn = self._ln_op(y, m + 'layer_norm')
        q = self.r(xn @ self.w[a + 'query_proj.linear.weight'].t() + self.w[a + 'query_proj.linear.bias']).view(B, T, h, dh)
        k = self.r(xn @ self.w[a + 'key_proj.linear.weight'].t() + self.w[a + 'key_proj.linear.bias']).view(B, T, h, dh)
        v = self.r(xn @ self.w[a + 'value_proj.linear.weight'].t() + self.w[a + 'value_proj.linear.bias']).view(B, T, h, dh)
        if T > self._maxlen:                                                   # embedding.py:35-41: the table is rebuilt for a longer input
            self._maxlen = T
            self._pe = sinusoid_table(hp.encoder_dim, T).to(self.dtype)
            self._ptab = {}
        P = self.pos_table(l).view(-1, h, dh)                                  # (2 max_len - 1, h, dh)
        cen = self._maxlen - 1
        i = torch.arange(T).view(T, 1)
        j = torch.arange(T).view(1, T)
        rel = cen - (i - j)                                                    # row index of P for (i,j)
        Pband = P[cen - (T - 1): cen + T]                                      # (2T-1,h,dh): p = T-1 ... -(T-1)
        gidx = (rel - (cen - (T - 1))).expand(B, h, T, T)
        if not self.bf16:
            qu = (q + self.w[a + 'u_bias']).permute(0, 2, 1, 3)                 # (B,h,T,dh)
            qv = (q + self.w[a + 'v_bias']).permute(0, 2, 1, 3)
            content = qu @ k.permute(0, 2, 3, 1)                                # (B,h,T,T)
            pos_all = qv @ Pband.permute(1, 2, 0)                               # (B,h,T,2T-1)
            pos = torch.gather(pos_all, 3, gidx)
            score = (content + pos) / math.sqrt(dh)
            attn = torch.softmax(score, -1)
            ctx = (attn @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, T, D)
        else:
            # the library folds log2(e)/sqrt(dh) into the query operands BEFORE rounding them (scores in log2 units,
            # 2^x softmax), rounds the softmax numerators to bf16 and sums those rounded values for the denominator
            sc = math.log2(math.e) / math.sqrt(dh)
            qu = self.r((q + self.w[a + 'u_bias']) * sc).permute(0, 2, 1, 3)
            qv = self.r((q + self.w[a + 'v_bias']) * sc).permute(0, 2, 1, 3)
            content = qu @ k.permute(0, 2, 3, 1)
            pos = torch.gather(qv @ Pband.permute(1, 2, 0), 3, gidx)
            s2 = content + pos
            pn = self.r(torch.exp2(s2 - s2.max(-1, keepdim=True).values))
            ctx = self.r((pn @ v.permute(0, 2, 1, 3)) / pn.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, T, D)
        out = ctx @ self.w[a + 'out_proj.linear.weight'].t() + self.w[a + 'out_proj.linear.bias']
        if taps is not None:
            taps[f'l{l}.q'] = q
            taps[f'l{l}.k'] = k
            taps[f'l{l}.v'] = v
            taps[f'l{l}.ctx'] = ctx
        return y + out

    def convmod(self, y: torch.Tensor, l: int, taps: Optional[dict] = None, glu: Optional[torch.Tensor] = None) -> torch.Tensor:
        """convolution.py:135-148 (SURVEY A.1b CONV): LN -> PW(D->2D) -> GLU -> depthwise k (zero pad at the
        PADDED batch edges) -> BatchNorm(eval, folded) -> SiLU -> PW(D->D); residual x1 (modules.py:32).
        `glu`: continue from this GLU output instead of the one computed here (tests that check the depthwise stage in isolation)."""
        hp = self.hp
        B, T, D = y.shape
        k = hp.conv_kernel_size
        c = f'encoder.layers.{l}.sequential.2.module.sequential.'
        t = self._ln_op(y, c + '0')
        a = t @ self.w[c + '2.conv.weight'].squeeze(-1).t() + self.w[c + '2.conv.bias']   # (B,T,2D)
        g = self.r(a[..., :D] * torch.sigmoid(a[..., D:]))
        if glu is not None:
            g = glu
        gp = F.pad(g, (0, 0, (k - 1) // 2, (k - 1) // 2))                      # zero rows before/after time
        if self.training:
            # nn.BatchNorm1d in train mode (convolution.py:141): statistics over ALL (batch, frame) positions of the padded batch
            # (no masking anywhere), biased variance for the normalisation; the caller updates the running statistics
            wraw = self.w[c + '4.conv.weight'].squeeze(1)
            dwo = sum(gp[:, tau:tau + T, :] * wraw[:, tau] for tau in range(k))
            mu = dwo.mean((0, 1))
            var = dwo.var((0, 1), unbiased=False)
            self.bn_batch_stats[l] = (mu.detach(), var.detach())
            u = (dwo - mu) / torch.sqrt(var + BN_EPS) * self.w[c + '5.weight'] + self.w[c + '5.bias']
        else:
            s = self.w[c + '5.weight'] / torch.sqrt(self.w[c + '5.running_var'] + BN_EPS)
            wdw = self.w[c + '4.conv.weight'].squeeze(1) * s.unsqueeze(1)      # (D,k) folded taps
            bdw = self.w[c + '5.bias'] - self.w[c + '5.running_mean'] * s
            u = bdw + sum(gp[:, tau:tau + T, :] * wdw[:, tau] for tau in range(k))
        u = self.r(F.silu(u))
        out = u @ self.w[c + '7.conv.weight'].squeeze(-1).t() + self.w[c + '7.conv.bias']
        if taps is not None:
            taps[f'l{l}.glu'] = g
            taps[f'l{l}.dw'] = u
        return y + out

    def block(self, y: torch.Tensor, l: int, taps: Optional[dict] = None) -> torch.Tensor:
        """encoder.py:67-100: FFN(half) -> MHSA -> Conv -> FFN(half) -> LayerNorm."""
        y = self.ffn(y, l, 0)
        if taps is not None:
            taps[f'l{l}.ffn1'] = y
        y = self.mhsa(y, l, taps)
        if taps is not None:
            taps[f'l{l}.mhsa'] = y
        y = self.convmod(y, l, taps)
        if taps is not None:
            taps[f'l{l}.conv'] = y
        y = self.ffn(y, l, 3)
        if taps is not None:
            taps[f'l{l}.ffn2'] = y
        y = self._ln(y, f'encoder.layers.{l}.sequential.4')
        if taps is not None:
            taps[f'l{l}.out'] = y
        return y

    # ------------------------------------------------------------------ whole path
    @torch.no_grad()
    def forward(self, line: torch.Tensor, lens, taps: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """PytorchRecognitionModel.forward (pred.py:118-122): line (N,1,H,W) -> (probits (N,T,ncls), lens int32).
        NO masking anywhere: the padded region takes part."""
        return self._forward(line, lens, taps)

    def forward_train(self, line: torch.Tensor, lens, taps: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The training forward of RecognitionModel._step (model.py:129-135): the same modules in train mode -- BatchNorm with batch
        statistics (`bn_batch_stats[l]` = (mean, biased variance) of block l, for the running-statistics update), dropout
        probabilities 0 -- with autograd enabled: set `requires_grad_` on the entries of `self.w` and call `.backward()` on a loss."""
        self.training = True
        try:
            return self._forward(line, lens, taps)
        finally:
            self.training = False

    def _forward(self, line: torch.Tensor, lens, taps: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        x = line.squeeze(1)                                                    # (N,H,W); the reference's transpose is a view
        y = self.frontend(x, taps)
        for l in range(self.hp.num_encoder_layers):
            y = self.block(y, l, taps)
        logits = self.r(y) @ self.w['decoder.weight'].t() + self.w['decoder.bias']      # (bf16 mode: the decoder reads the bf16 `xn`)
        if taps is not None:
            taps['logits'] = logits
        return logits, out_len(lens, self.hp.sampling_num)
