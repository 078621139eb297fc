"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference S3TokenizerV2/V3 (fp32, torch CPU),
following the padded-batch + mask arithmetic of the Swift literally.

parity unpinned: no golden vectors in the reference.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Follows (paths relative to /root/reference/package/Codec/S3Tokenizer):
  S3Tokenizer.swift:13-68 (RoPE table: freqs = theta^-(j/dim), j < dim/2; rotation [-x_R, x_L]), :149-168 (FSQ),
  :225-315 (FSMN attention), :321-354 (block), :396-436 (AudioEncoderV2), :474-494 (quantize), :497-650 (long-audio windows)
  S3TokenizerUtils.swift:21-41 (masks), :71-88 (mergeTokenizedSegments)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

torch.set_grad_enabled(False)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


def _conv1d_cl(x, w, b, stride, padding, groups=1):
    """MLX Conv1d, channels-last [B,L,C], weight [Cout,K,Cin/groups]."""
    return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), b, stride=stride, padding=padding, groups=groups).transpose(1, 2)


class S3Oracle:
    def __init__(self, cfg, weights):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in weights.items()}
        half = 32
        freqs = 1.0 / torch.pow(torch.tensor(10000.0), torch.arange(half, dtype=torch.float32) / 64.0)   # S3Tokenizer.swift:19-21
        t = torch.arange(2048, dtype=torch.float32)
        fo = torch.outer(t, freqs)
        self.cos = torch.cat([torch.cos(fo), torch.cos(fo)], dim=-1)
        self.sin = torch.cat([torch.sin(fo), torch.sin(fo)], dim=-1)

    def _block(self, p, x, mask_bias, mask_pad):
        W, H = self.w, self.cfg.n_audio_head
        B, T, D = x.shape
        h = F.layer_norm(x, (D,), W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"], 1e-5)
        q = h @ W[p + ".attn.query.weight"].t() + W[p + ".attn.query.bias"]
        k = h @ W[p + ".attn.key.weight"].t()
        v = h @ W[p + ".attn.value.weight"].t() + W[p + ".attn.value.bias"]
        scale = float(D // H) ** -0.25
        qr, kr, vr = q.reshape(B, T, H, -1), k.reshape(B, T, H, -1), v.reshape(B, T, H, -1)
        cos, sin = self.cos[:T][None, :, None, :], self.sin[:T][None, :, None, :]
        rot = lambda z: torch.cat([-z[..., 32:], z[..., :32]], dim=-1)
        qr, kr = qr * cos + rot(qr) * sin, kr * cos + rot(kr) * sin
        # FSMN memory on V (S3Tokenizer.swift:225-251)
        vi = vr.reshape(B, T, D) * mask_pad
        mem = _conv1d_cl(F.pad(vi, (0, 0, 15, 15)), W[p + ".attn.fsmn_block.weight"], None, 1, 0, groups=D) + vi
        mem = mem * mask_pad
        qt, kt, vt = qr.transpose(1, 2) * scale, kr.transpose(1, 2) * scale, vr.transpose(1, 2)
        s = qt @ kt.transpose(-1, -2) + mask_bias[:, None, :, :]
        o = (torch.softmax(s, dim=-1) @ vt).transpose(1, 2).reshape(B, T, D)
        x = x + (o @ W[p + ".attn.out.weight"].t() + W[p + ".attn.out.bias"] + mem)
        h = F.layer_norm(x, (D,), W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"], 1e-5)
        g = F.gelu(h @ W[p + ".mlp.layers.0.weight"].t() + W[p + ".mlp.layers.0.bias"])
        return x + g @ W[p + ".mlp.layers.2.weight"].t() + W[p + ".mlp.layers.2.bias"]

    def quantize(self, mel: np.ndarray, mel_len: np.ndarray):
        """mel [B, n_mels, T], mel_len [B] (all <= 3000 frames) -> (codes int32 [B, T''], code_len [B], pre-round FSQ values [B, T'', 8])."""
        W = self.w
        x = _t(mel)
        lens = torch.as_tensor(np.asarray(mel_len, np.int64))
        T = x.shape[2]
        nonpad = lambda l, n: (torch.arange(n)[None, :] < l[:, None]).float()
        x = x.transpose(1, 2) * nonpad(lens, T)[:, :, None]
        x = F.gelu(_conv1d_cl(x, W["encoder.conv1.weight"], W["encoder.conv1.bias"], 2, 1))
        lens = (lens + 2 - 2 - 1) // 2 + 1
        T = (T + 2 - 2 - 1) // 2 + 1
        x = F.gelu(_conv1d_cl(x * nonpad(lens, T)[:, :, None], W["encoder.conv2.weight"], W["encoder.conv2.bias"], 2, 1))
        lens = (lens + 2 - 2 - 1) // 2 + 1
        T = (T + 2 - 2 - 1) // 2 + 1
        m = nonpad(lens, T)
        mask_pad = m[:, :, None]
        mask_bias = ((1.0 - m) * -1.0e10)[:, None, :]
        for l in range(self.cfg.n_audio_layer):
            x = self._block(f"encoder.blocks.{l}", x, mask_bias, mask_pad)
        pre = torch.tanh(x @ W["quantizer.fsq_codebook.project_down.weight"].t() + W["quantizer.fsq_codebook.project_down.bias"]) * 0.9990000128746033
        h = torch.round(pre) + 1                                          # torch.round is half-to-even like MLX
        powers = torch.pow(torch.tensor(3.0), torch.arange(8, dtype=torch.float32))
        ids = torch.sum(h * powers[None, None, :], dim=-1).to(torch.int32)
        return ids.numpy(), lens.numpy().astype(np.int32), pre.numpy()     # pre: the PRE-round FSQ values (tests prove boundary cases with them)


def merge_tokenized_segments(segments: list[list[int]], overlap: int = 4, token_rate: int = 25) -> list[int]:
    """S3TokenizerUtils.swift:71-88."""
    out: list[int] = []
    ot = (overlap // 2) * token_rate
    for i, toks in enumerate(segments):
        left = 0 if i == 0 else ot
        right = len(toks) - ot if i != len(segments) - 1 else len(toks)
        if left < right:
            out.extend(toks[left:right])
    return out
