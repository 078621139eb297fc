"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference SNAC and DAC decoders (fp32, torch CPU).

parity unpinned: the reference holds no golden vectors for the codecs and the MLX primitives it
calls (conv1d / convTransposed1d, channels-last, weight [Cout,K,Cin]) are restated from the public MLX documentation:
conv_transpose1d is taken to be the gradient-of-convolution form y[t*s + k - p, co] += x[t, ci] * w[co, k, ci]
(== torch.conv_transpose1d with weight[ci, co, k]).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to /root/reference/package):
  TTS/Orpheus/SNAC/SNACDecoder.swift:250-289 (forward/decode), :328-407 (embedCodes), :409-416 (snake), :466-489 (block)
  TTS/Orpheus/SNAC/WNConv1d.swift:64-88, ConvWeightedTranspose1d.swift:70-100, ResidualUnit.swift:58-95, NoiseBlock.swift:27-41
  Codec/DAC/DACLayers.swift:14-30 (norm, snake), :92-117, :167-192, :198-235; DACQuantize.swift:192-220; DACModel.swift:120-164,303-306
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

torch.set_grad_enabled(False)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


def _norm_except(v: torch.Tensor, except_dim: int) -> torch.Tensor:
    axes = [a for a in range(v.ndim) if a != except_dim]
    return torch.sqrt(torch.sum(v * v, dim=axes, keepdim=True))


def _snake(x, alpha):
    """x [B,C,T], alpha broadcastable [1,C,1]: x + sin^2(alpha x) / (alpha + 1e-9)."""
    return x + (1.0 / (alpha + 1e-9)) * torch.sin(alpha * x) ** 2


def _conv1d_cf(x, w_mlx, b, stride=1, padding=0, dilation=1, groups=1):
    """x [B,C,T]; w_mlx [Cout,K,Cin/groups] (MLX layout)."""
    return F.conv1d(x, w_mlx.permute(0, 2, 1).contiguous(), b, stride=stride, padding=padding, dilation=dilation, groups=groups)


def _convt1d_cf(x, w_mlx, b, stride, padding):
    """x [B,Cin,T]; w_mlx [Cout,K,Cin] (MLX layout); no output_padding (neither port passes it)."""
    return F.conv_transpose1d(x, w_mlx.permute(2, 0, 1).contiguous(), b, stride=stride, padding=padding)


class SNACOracle:
    def __init__(self, cfg, weights: dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in weights.items()}

    def _wnconv(self, p, x, padding=0, dilation=1, groups=1, bias=True):
        W = self.w
        v, g = W[p + ".weight_v"], W[p + ".weight_g"]
        weight = g * v / (_norm_except(v, 0) + 1e-12)                      # WNConv1d.swift:73-74
        return _conv1d_cf(x, weight, W[p + ".bias"] if bias else None, padding=padding, dilation=dilation, groups=groups)

    def _wnconvt(self, p, x, stride):
        W = self.w
        v, g = W[p + ".weight_v"], W[p + ".weight_g"]                      # v [Cin,K,Cout], g [Cin,1,1]
        eff = (g * v) / (torch.sqrt(torch.sum(v * v, dim=(1, 2), keepdim=True)) + 1e-12)
        weight = eff.permute(2, 1, 0).contiguous()                         # [Cout,K,Cin]  (ConvWeightedTranspose1d.swift:82-84)
        pad = int(np.ceil(stride / 2.0))
        return _convt1d_cf(x, weight, W[p + ".bias"], stride, pad)

    def embed_codes(self, codes: list[list[int]]) -> torch.Tensor:
        """SNACDecoder.swift:328-407 -> [latent, T] (un-batched)."""
        cfg, W = self.cfg, self.w
        T = 0
        for i, s in enumerate(cfg.vq_strides):
            if i < len(codes) and len(codes[i]):
                T = max(T, len(codes[i]) * s)
        z = torch.zeros(cfg.latent_dim, T)
        for i, s in enumerate(cfg.vq_strides):
            if i >= len(codes) or not len(codes[i]):
                continue
            q = f"quantizer.quantizers.{i}"
            dec = W[q + ".codebook.weight"][torch.as_tensor(codes[i], dtype=torch.long)]
            g = W[q + ".out_proj.weight_g"].reshape(-1)
            v = W[q + ".out_proj.weight_v"].reshape(cfg.latent_dim, -1)
            eff = g.reshape(-1, 1) * v / (torch.sqrt(torch.sum(v * v, dim=1, keepdim=True)) + 1e-12)
            proj = (dec @ eff.t() + W[q + ".out_proj.bias"]).t()           # [latent, n_i]
            exp = proj.repeat_interleave(s, dim=1) if s > 1 else proj
            if exp.shape == z.shape:
                z = z + exp
        return z

    def decode(self, codes: list[list[int]], noise: np.ndarray | None = None) -> np.ndarray:
        """SNACDecoder.decode(codes:) -> float32 [samples].  `noise`: concatenated N(0,1) draws per NoiseBlock, or None (= 0)."""
        cfg, W = self.cfg, self.w
        P = "decoder.model.layers."
        y = self.embed_codes(codes)[None]                                   # [1, C, T]
        y = self._wnconv(P + "0", y, padding=3, groups=cfg.latent_dim)
        y = self._wnconv(P + "1", y)
        noff = 0
        for i, s in enumerate(cfg.decoder_rates):
            b = f"{P}{2 + i}.block.layers."
            cout = cfg.decoder_dim >> (i + 1)
            y = _snake(y, W[b + "0.alpha"])
            y = self._wnconvt(b + "1", y, s)
            ru = 2
            if cfg.noise:
                T = y.shape[2]
                h = self._wnconv(b + "2.linear", y, bias=False)              # [1, Cn, T]
                if noise is not None:
                    nz = _t(noise[noff:noff + T]).reshape(1, 1, T)
                    y = y + nz * h                                           # NoiseBlock.swift:33-41
                noff += T
                ru = 3
            for r, d in enumerate((1, 3, 9)):
                u = f"{b}{ru + r}.block.layers."
                res = y
                t = _snake(y, W[u + "0.alpha"])
                t = self._wnconv(u[:-1] + ".1", t, padding=3 * d, dilation=d, groups=cout)
                t = _snake(t, W[u + "2.alpha"])
                t = self._wnconv(u[:-1] + ".3", t)
                y = res + t
        n = len(cfg.decoder_rates)
        y = _snake(y, W[f"{P}{2 + n}.alpha"])
        y = self._wnconv(f"{P}{3 + n}", y, padding=3)
        return torch.tanh(y).reshape(-1).numpy()

    def noise_len(self, latent_len: int) -> int:
        T, total = latent_len, 0
        for s in self.cfg.decoder_rates:
            T = (T - 1) * s - 2 * int(np.ceil(s / 2.0)) + 2 * s
            if self.cfg.noise:
                total += T
        return total


class DACOracle:
    def __init__(self, cfg, weights: dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in weights.items()}

    def _wnconv(self, p, x, padding=0, dilation=1, stride=1):
        W = self.w
        v, g = W[p + ".weight_v"], W[p + ".weight_g"]
        weight = g * v / (_norm_except(v, 0) + 1e-12)                      # DACLayers.swift:106-107
        return _conv1d_cf(x, weight, W[p + ".bias"], stride=stride, padding=padding, dilation=dilation)

    def _res_unit(self, u, y, d):
        """DACResidualUnit (DACLayers.swift:198-235): snake -> conv7 dilated -> snake -> conv1 -> + x."""
        W = self.w
        t = _snake(y, W[u + "0.alpha"].reshape(1, -1, 1))
        t = self._wnconv(u[:-1] + ".1", t, padding=3 * d, dilation=d)
        t = _snake(t, W[u + "2.alpha"].reshape(1, -1, 1))
        return y + self._wnconv(u[:-1] + ".3", t)

    def encode(self, audio: np.ndarray, n_quantizers: int | None = None):
        """DACCodec.encode (DACModel.swift:284-296) for one mono sequence: preprocess (right-pad to the hop length, :308-317) ->
        DACEncoder (:43-86, blocks :15-38) -> residual vector quantisation (DACQuantize.swift:147-190, nearest entry on L2-normalised
        vectors :87-115).  Returns (codes int32 [n_q, T'], gaps float32 [n_q, T'] = second-best minus best distance of every
        choice: oracle-only diagnostic, a 16-byte-different matmul may pick the runner-up only where the gap is ~0)."""
        cfg, W = self.cfg, self.w
        hop = int(np.prod(cfg.encoder_rates))
        x = np.asarray(audio, np.float32).reshape(-1)
        pad = -(-x.shape[0] // hop) * hop - x.shape[0]
        y = _t(np.pad(x, (0, pad)))[None, None, :]
        E = "encoder.block.layers."
        y = self._wnconv(E + "0", y, padding=3)
        for i, st in enumerate(cfg.encoder_rates):
            b = f"{E}{1 + i}.block.layers."
            for r, d in enumerate((1, 3, 9)):
                y = self._res_unit(f"{b}{r}.block.layers.", y, d)
            y = _snake(y, W[b + "3.alpha"].reshape(1, -1, 1))
            y = self._wnconv(b + "4", y, padding=int(np.ceil(st / 2.0)), stride=st)
        ne = len(cfg.encoder_rates)
        y = _snake(y, W[f"{E}{1 + ne}.alpha"].reshape(1, -1, 1))
        z = self._wnconv(f"{E}{2 + ne}", y, padding=1)                     # [1, latent, T']

        def l2n(a):                                                        # l2Normalize (DACQuantize.swift:14-20), dim 1
            n = torch.pow(torch.sum(torch.pow(torch.abs(a), 2), dim=1, keepdim=True), 0.5)
            return a / torch.maximum(n, torch.tensor(1e-12))

        nq = cfg.n_codebooks if n_quantizers is None else min(n_quantizers, cfg.n_codebooks)
        residual, codes, gaps = z, [], []
        for i in range(nq):
            q = f"quantizer.quantizers.{i}"
            zE = self._wnconv(q + ".in_proj", residual)                   # [1, cb_dim, T']
            enc = zE[0].t()
            cb = W[q + ".codebook.weight"]
            en, cn = l2n(enc), l2n(cb)
            dist = torch.sum(en ** 2, dim=1, keepdim=True) - 2 * en @ cn.t() + torch.sum(cn ** 2, dim=1, keepdim=True).t()
            idx = torch.argmax(-dist, dim=1)
            two = torch.topk(-dist, 2, dim=1).values
            gaps.append((two[:, 0] - two[:, 1]).numpy())
            zq_raw = cb[idx].t()[None]
            zq = zE + (zq_raw - zE)                                        # straight-through form of the forward pass (:62)
            residual = residual - self._wnconv(q + ".out_proj", zq)
            codes.append(idx.numpy().astype(np.int32))
        return np.stack(codes), np.stack(gaps).astype(np.float32)

    def decode_from_codes(self, codes: np.ndarray) -> np.ndarray:
        """DACCodec.decodeFromCodes: codes int [n_codebooks, T] (one sequence) -> float32 [samples]."""
        cfg, W = self.cfg, self.w
        z = None
        for i in range(codes.shape[0]):
            q = f"quantizer.quantizers.{i}"
            zp = W[q + ".codebook.weight"][torch.as_tensor(codes[i], dtype=torch.long)].t()[None]      # [1, 8, T]
            zq = self._wnconv(q + ".out_proj", zp)
            z = zq if z is None else z + zq                                   # DACQuantize.swift:192-220
        P = "decoder.model.layers."
        y = self._wnconv(P + "0", z, padding=3)
        for i, s in enumerate(cfg.decoder_rates):
            b = f"{P}{1 + i}.block.layers."
            y = _snake(y, W[b + "0.alpha"].reshape(1, -1, 1))
            v, g = W[b + "1.weight_v"], W[b + "1.weight_g"]                 # [Cout,K,Cin], [1,1,Cin]
            weight = g * v / (_norm_except(v, 2) + 1e-12)                    # DACLayers.swift:180-181
            y = _convt1d_cf(y, weight, W[b + "1.bias"], s, int(np.ceil(s / 2.0)))
            for r, d in enumerate((1, 3, 9)):
                y = self._res_unit(f"{b}{2 + r}.block.layers.", y, d)
        n = len(cfg.decoder_rates)
        y = _snake(y, W[f"{P}{1 + n}.alpha"].reshape(1, -1, 1))
        y = self._wnconv(f"{P}{2 + n}", y, padding=3)
        return torch.tanh(y).reshape(-1).numpy()
