"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float32 + torch fp32 convolutions) of CosyVoice2's HiFT vocoder.

Follows, as text, TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift (linearInterpolate1d :17-60, SineGen2 :66-154,
SourceModuleHnNSF2 :160-204, CosyF0Predictor :212-256, CosyHiFTGenerator.decode :412-475, callAsFunction :482-505) and
Codec/S3Gen/HiFiGAN.swift (hannWindowPeriodic :15-20, Snake :30-67, HiFiGANResBlock :75-130, stftHiFiGAN :257-295,
istftHiFiGAN :298-367).

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be run here (Swift + MLX + Metal); this file is
pinned only by construction (line-by-line restatement) and by the self-consistency checks in tests/test_oracle_hift.py (STFT ->
iSTFT round trip against numpy.fft, interpolation against torch.nn.functional.interpolate).  The product never imports it.
The reference's two random draws (initial phases, Gaussian source noise) are explicit arguments.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


def hann_periodic(n: int) -> np.ndarray:
    return np.array([0.5 * (1 - np.cos(2 * f32(np.pi) * f32(i) / f32(n))) for i in range(n)], dtype=f32)


def linear_interpolate_1d(x: np.ndarray, scale: np.float32) -> np.ndarray:
    """linearInterpolate1d (:17-60) on [T, C] float32, every operation rounded to float32 in the reference's order."""
    T = x.shape[0]
    new_t = int(f32(T) * f32(scale)) or 1
    idx = (np.arange(new_t, dtype=f32) + f32(0.5)) * (f32(T) / f32(new_t)) - f32(0.5)
    idx = np.minimum(np.maximum(idx, f32(0)), f32(T) - f32(1.001))
    lo = np.floor(idx).astype(np.int32)
    hi = np.minimum(lo + 1, T - 1)
    wh = idx - lo.astype(f32)
    wl = f32(1) - wh
    return (x[lo] * wl[:, None] + x[hi] * wh[:, None]).astype(f32)


def sine_gen2(f0_up: np.ndarray, cfg, rand_ini: np.ndarray | None, noise: np.ndarray | None) -> np.ndarray:
    """SineGen2.callAsFunction (:134-153): f0_up [L] -> sine waves [L, H]."""
    H, up = cfg.nb_harmonics + 1, cfg.upsample_factor
    L = f0_up.shape[0]
    fn = f0_up.astype(f32)[:, None] * np.arange(1, H + 1, dtype=f32)[None, :]
    rad = np.fmod(fn / f32(cfg.sampling_rate), f32(1)).astype(f32)
    if rand_ini is not None:
        ini = rand_ini.astype(f32).copy(); ini[0] = 0
        rad[0] = rad[0] + ini
    rad_d = linear_interpolate_1d(rad, f32(1.0) / f32(up))
    phase = np.cumsum(rad_d, axis=0, dtype=f32) * f32(2) * f32(np.pi)
    phase = linear_interpolate_1d((phase * f32(up)).astype(f32), f32(up))[:L]
    sine = (np.sin(phase) * f32(cfg.nsf_alpha)).astype(f32)
    uv = (f0_up > f32(cfg.voiced_threshold)).astype(f32)[:, None]
    out = sine * uv
    if noise is not None:
        namp = uv * f32(cfg.nsf_sigma) + (f32(1) - uv) * f32(cfg.nsf_alpha) / f32(3)
        out = out + namp * noise.astype(f32)
    return out.astype(f32)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _conv(w, p, x, stride=1, padding=0, dilation=1):
    """MLX Conv1d on [1, C, T] with weight [Cout, K, Cin]."""
    return F.conv1d(x, _t(w[p + ".weight"]).permute(0, 2, 1), _t(w[p + ".bias"]), stride=stride, padding=padding, dilation=dilation)


def snake(x, alpha):
    a = _t(alpha).reshape(1, -1, 1)
    ab = a.abs()
    cl = torch.sign(a) * torch.clamp(ab, min=1e-4)
    cl = torch.where(ab < 1e-9, torch.full_like(a, 1e-4), cl)
    return x + (1.0 / cl) * torch.sin(x * a) ** 2


def resblock(w, p, x, k, dilations):
    res = x
    for i, d in enumerate(dilations):
        xt = snake(res, w[f"{p}.activations1.{i}.alpha"])
        xt = _conv(w, f"{p}.convs1.{i}", xt, padding=(k * d - d) // 2, dilation=d)
        xt = snake(xt, w[f"{p}.activations2.{i}.alpha"])
        xt = _conv(w, f"{p}.convs2.{i}", xt, padding=(k - 1) // 2)
        res = xt + res
    return res


def f0_predictor(w, mel: np.ndarray) -> np.ndarray:
    """CosyF0Predictor (:236-255): mel [C, T] -> f0 [T]."""
    h = _t(mel)[None]
    for i in (0, 2, 4, 6, 8):
        h = F.elu(_conv(w, f"f0_predictor.condnet_{i}", h, padding=1))
    f0 = F.linear(h.transpose(1, 2), _t(w["f0_predictor.classifier.weight"]), _t(w["f0_predictor.classifier.bias"]))[0, :, 0]
    return f0.abs().numpy()


def source(w, cfg, f0: np.ndarray, noise: np.ndarray | None, rand_ini: np.ndarray | None = None) -> np.ndarray:
    """f0Upsample + SourceModuleHnNSF2 (:408-410, :190-203): f0 [T] -> s [L]."""
    f0_up = np.repeat(f0.astype(f32), cfg.upsample_factor)
    sw = sine_gen2(f0_up, cfg, rand_ini, noise)
    lin = sw @ w["m_source.l_linear.weight"].astype(f32).T + w["m_source.l_linear.bias"].astype(f32)
    return np.tanh(lin[:, 0]).astype(f32)


def stft(x: np.ndarray, n_fft: int = 16, hop: int = 4):
    """stftHiFiGAN (:257-295): x [L] -> (real, imag) [n_fft/2+1, F]."""
    pad = n_fft // 2
    xp = np.concatenate([x[1:pad + 1][::-1], x, x[-pad - 1:-1][::-1]]).astype(f32)
    nfr = (xp.shape[0] - n_fft) // hop + 1
    fr = np.stack([xp[i * hop:i * hop + n_fft] for i in range(nfr)], axis=1) * hann_periodic(n_fft)[:, None]
    sp = np.fft.fft(fr.astype(np.float32), axis=0)[:n_fft // 2 + 1]
    return sp.real.astype(f32), sp.imag.astype(f32)


def istft(mag: np.ndarray, phase: np.ndarray, n_fft: int = 16, hop: int = 4) -> np.ndarray:
    """istftHiFiGAN (:298-367): [9, F] magnitude / phase -> [4 (F - 1)]."""
    mag = np.minimum(mag, f32(1e2))
    re, im = mag * np.cos(phase), mag * np.sin(phase)
    re_full = np.concatenate([re, re[1:-1][::-1]], axis=0)
    im_full = np.concatenate([im, -im[1:-1][::-1]], axis=0)
    fr = np.fft.ifft(re_full + 1j * im_full, axis=0).real.astype(f32) * hann_periodic(n_fft)[:, None]
    nfr = fr.shape[1]
    out_len = (nfr - 1) * hop + n_fft
    idx = (np.arange(nfr)[:, None] * hop + np.arange(n_fft)[None, :]).reshape(-1)
    wsum = np.zeros(out_len, f32)
    np.add.at(wsum, idx, np.tile(hann_periodic(n_fft) ** 2, nfr))
    wsum = np.maximum(wsum, f32(1e-8))
    out = np.zeros(out_len, f32)
    np.add.at(out, idx, fr.T.reshape(-1))
    out = out / wsum
    return out[n_fft // 2:out_len - n_fft // 2].astype(f32)


def decode(w, cfg, mel: np.ndarray, s: np.ndarray) -> np.ndarray:
    """CosyHiFTGenerator.decode (:412-475): mel [C, T], s [L] -> waveform [L]."""
    sr, si = stft(s, cfg.n_fft, cfg.hop)
    s_stft = _t(np.concatenate([sr, si], axis=0))[None]
    h = _conv(w, "conv_pre", _t(mel)[None], padding=3)
    n = len(cfg.up_rates)
    nk = len(cfg.res_kernels)
    for i in range(n):
        h = F.leaky_relu(h, cfg.lrelu_slope)
        k, u = cfg.up_kernels[i], cfg.up_rates[i]
        h = F.conv_transpose1d(h, _t(w[f"ups.{i}.weight"]).permute(2, 0, 1), _t(w[f"ups.{i}.bias"]), stride=u, padding=(k - u) // 2)
        if i == n - 1:
            h = torch.cat([h[:, :, 1:2], h], dim=2)
        dr = int(np.prod(cfg.up_rates[i + 1:])) if i + 1 < n else 1
        sd = _conv(w, f"source_downs.{i}", s_stft) if dr == 1 else _conv(w, f"source_downs.{i}", s_stft, stride=dr, padding=dr // 2)
        sd = resblock(w, f"source_resblocks.{i}", sd, cfg.src_res_kernels[i], cfg.dilations)
        h = h + sd
        outs = [resblock(w, f"resblocks.{i * nk + j}", h, cfg.res_kernels[j], cfg.dilations) for j in range(nk)]
        acc = outs[0]
        for o in outs[1:]:
            acc = acc + o
        h = acc / float(nk)
    h = F.leaky_relu(h, 0.01)
    h = _conv(w, "conv_post", h, padding=3)[0].numpy()
    nb = cfg.n_fft // 2 + 1
    out = istft(np.exp(h[:nb]), np.sin(h[nb:]), cfg.n_fft, cfg.hop)
    return np.clip(out, -f32(cfg.audio_limit), f32(cfg.audio_limit)).astype(f32)


def vocode(w, cfg, mel: np.ndarray, noise: np.ndarray | None = None, cache_source: np.ndarray | None = None, rand_ini=None):
    """callAsFunction (:482-505) -> (waveform [L], source [L])."""
    f0 = f0_predictor(w, mel)
    s = source(w, cfg, f0, noise, rand_ini)
    if cache_source is not None and cache_source.shape[0]:
        s = np.concatenate([cache_source.astype(f32), s[cache_source.shape[0]:]])
    return decode(w, cfg, mel, s), s
