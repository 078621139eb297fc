"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float32 + torch fp32 convolutions) of the CAM++ speaker encoder and its
Kaldi-style filterbank front end, the one-off-per-speaker part of CosyVoice2's prepareConditionals.

Follows, as text, Codec/S3Gen/CAMPPlus.swift: poveyWindow :15-20, kaldiFbankCAMPPlus :32-108, computeMelFiltersHTK :134-170,
BasicResBlock :180-242, FCM :246-324, statisticsPooling :328-333, TDNNLayer :345-393, CAMLayer :420-503 (segPooling :470-493),
CAMDenseTDNNLayer :507-567, CAMDenseTDNNBlock :571-609, TransitLayer :613-638, DenseLayer :642-683, CAMPPlus :687-785,
inference :788-818; and TTS/CosyVoice2/SpeakerEncoder/CAMPlusSpeakerEncoder.swift:12-150 (configuration, zero embedding when no
weights are loaded).

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be run here (Swift + MLX + Metal); this file is
pinned only by construction and by the self-consistency checks in tests/test_oracle_campplus.py.  BatchNorm runs in inference
mode (running statistics, eps 1e-5 = the MLXNN default).  The product never imports it.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32
BN_EPS = 1e-5


# ---- Kaldi filterbank (CAMPPlus.swift:15-170) --------------------------------------------------------------------------------------
def povey_window(size: int) -> np.ndarray:
    n = np.arange(size, dtype=f32)
    hann = f32(0.5) - f32(0.5) * np.cos(f32(2) * f32(np.pi) * n / f32(size - 1)).astype(f32)
    return np.power(hann, f32(0.85)).astype(f32)


def next_power_of_2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


def mel_filters_htk(sample_rate: int, n_fft: int, n_mels: int, f_min: float, f_max: float) -> np.ndarray:
    """computeMelFiltersHTK (:134-170): triangles on ROUNDED FFT-bin edges, [n_fft/2+1, n_mels]."""
    def hz_to_mel(hz):
        return f32(2595.0) * f32(math.log10(f32(1.0) + f32(hz) / f32(700.0)))

    def mel_to_hz(mel):
        return f32(700.0) * (f32(math.pow(10.0, f32(mel) / f32(2595.0))) - f32(1.0))

    mel_min, mel_max = hz_to_mel(f_min), hz_to_mel(f_max)
    mel_points = [f32(mel_min + f32(i) * (mel_max - mel_min) / f32(n_mels + 1)) for i in range(n_mels + 2)]
    hz_points = [mel_to_hz(m) for m in mel_points]

    def swift_round(v):                                   # Foundation round(): half away from zero
        return int(math.floor(float(v) + 0.5)) if v >= 0 else -int(math.floor(-float(v) + 0.5))

    bins = [swift_round(f32(h) * f32(n_fft) / f32(sample_rate)) for h in hz_points]
    nb = n_fft // 2 + 1
    filt = np.zeros((nb, n_mels), f32)
    for m in range(1, n_mels + 1):
        lo, c, hi = bins[m - 1], bins[m], bins[m + 1]
        if c != lo:
            for k in range(lo, c):
                if 0 <= k < nb:
                    filt[k, m - 1] = f32(k - lo) / f32(c - lo)
        if hi != c:
            for k in range(c, hi):
                if 0 <= k < nb:
                    filt[k, m - 1] = f32(hi - k) / f32(hi - c)
    return filt


def kaldi_fbank(audio: np.ndarray, sample_rate: int = 16000, num_mel_bins: int = 80, frame_length: float = 25.0,
                frame_shift: float = 10.0) -> np.ndarray:
    """kaldiFbankCAMPPlus (:32-108): [n] -> [frames, 80] log mel energies (snip_edges framing, DC removal, 0.97 pre-emphasis inside
    the frame, Povey window, 512-point power spectrum, HTK triangles, log(max(., 1.1920929e-07)))."""
    win = int(f32(sample_rate) * f32(frame_length) / f32(1000))
    hop = int(f32(sample_rate) * f32(frame_shift) / f32(1000))
    n_fft = next_power_of_2(win)
    x = np.asarray(audio, f32).reshape(-1)
    n_frames = max((x.shape[0] - win) // hop + 1, 1)
    idx = (np.arange(n_frames)[:, None] * hop + np.arange(win)[None, :])
    frames = x[idx].astype(f32)
    frames = frames - frames.mean(axis=1, keepdims=True, dtype=f32)
    frames = np.concatenate([frames[:, :1], frames[:, 1:] - f32(0.97) * frames[:, :-1]], axis=1).astype(f32)
    frames = frames * povey_window(win)[None, :]
    frames = np.concatenate([frames, np.zeros((n_frames, n_fft - win), f32)], axis=1)
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    power = (np.abs(spec) ** 2).astype(f32)
    mel = (power @ mel_filters_htk(sample_rate, n_fft, num_mel_bins, 20.0, sample_rate / 2)).astype(f32)
    return np.log(np.maximum(mel, f32(1.1920929e-07))).astype(f32)


# ---- network (CAMPPlus.swift:180-785) --------------------------------------------------------------------------------------------------
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=f32))


class CAMPPlusOracle:
    """weights: flat dict with the reference's Module key paths (MLX layouts: Conv2d [O, KH, KW, I], Conv1d [O, K, I])."""

    def __init__(self, weights: dict[str, np.ndarray], feat_dim: int = 80):
        self.w = weights
        self.feat_dim = feat_dim

    # BatchNorm, inference mode, over the channel axis `dim` of a torch tensor
    def bn(self, x, key, dim, affine=True):
        rm, rv = _t(self.w[key + ".running_mean"]), _t(self.w[key + ".running_var"])
        shape = [1] * x.ndim
        shape[dim] = -1
        y = (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + BN_EPS)
        if affine:
            y = y * _t(self.w[key + ".weight"]).view(shape) + _t(self.w[key + ".bias"]).view(shape)
        return y

    def conv2d(self, x, key, stride, pad):               # x [B, C, H, W]
        w = _t(self.w[key + ".weight"]).permute(0, 3, 1, 2)
        return F.conv2d(x, w, None, stride=stride, padding=pad)

    def conv1d(self, x, key, stride=1, pad=0, dil=1, bias=False):   # x [B, C, T]
        w = _t(self.w[key + ".weight"]).permute(0, 2, 1)
        b = _t(self.w[key + ".bias"]) if bias else None
        return F.conv1d(x, w, b, stride=stride, padding=pad, dilation=dil)

    def res_block(self, x, p, stride):
        out = torch.relu(self.bn(self.conv2d(x, p + ".conv1", (stride, 1), 1), p + ".bn1", 1))
        out = self.bn(self.conv2d(out, p + ".conv2", 1, 1), p + ".bn2", 1)
        sc = x
        if (p + ".shortcut.0.weight") in self.w:
            sc = self.bn(self.conv2d(x, p + ".shortcut.0", (stride, 1), 0), p + ".shortcut.1", 1)
        return torch.relu(out + sc)

    def fcm(self, x):                                    # x [B, F, T] -> [B, 32 * F/8, T]
        out = x[:, None]                                 # [B, 1, H = F, W = T]
        out = torch.relu(self.bn(self.conv2d(out, "head.conv1", 1, 1), "head.bn1", 1))
        out = self.res_block(out, "head.layer1.0", 2)
        out = self.res_block(out, "head.layer1.1", 1)
        out = self.res_block(out, "head.layer2.0", 2)
        out = self.res_block(out, "head.layer2.1", 1)
        out = torch.relu(self.bn(self.conv2d(out, "head.conv2", (2, 1), 1), "head.bn2", 1))
        B, C, H, W = out.shape
        return out.reshape(B, C * H, W)                  # channel index c * H + h

    @staticmethod
    def seg_pooling(x, seg_len=100):                     # [B, C, T] -> [B, C, T]; the padded tail divides by seg_len too
        B, C, T = x.shape
        n = (T + seg_len - 1) // seg_len
        xp = F.pad(x, (0, n * seg_len - T))
        seg = xp.reshape(B, C, n, seg_len).mean(dim=-1, keepdim=True).expand(B, C, n, seg_len).reshape(B, C, -1)
        return seg[:, :, :T]

    def cam_layer(self, x, p, dil):
        y = self.conv1d(x, p + ".linear_local", 1, dil, dil)
        ctx = x.mean(dim=-1, keepdim=True) + self.seg_pooling(x)
        ctx = torch.relu(self.conv1d(ctx, p + ".linear1", bias=True))
        m = torch.sigmoid(self.conv1d(ctx, p + ".linear2", bias=True))
        return y * m

    def dense_layer(self, x, p, dil):
        out = torch.relu(self.bn(x, p + ".nonlinear1.0", 1))
        out = self.conv1d(out, p + ".linear1")
        out = torch.relu(self.bn(out, p + ".nonlinear2.0", 1))
        return self.cam_layer(out, p + ".cam_layer", dil)

    def forward(self, feats: np.ndarray) -> np.ndarray:
        """feats [B, T, F] (mean-normalised fbank) -> [B, 192]."""
        with torch.no_grad():
            out = self.fcm(_t(feats).transpose(1, 2))
            out = torch.relu(self.bn(self.conv1d(out, "tdnn.linear", 2, 2, 1), "tdnn.nonlinear.0", 1))
            for b, (n_layers, dil) in enumerate([(12, 1), (24, 2), (16, 2)]):
                for i in range(n_layers):
                    out = torch.cat([out, self.dense_layer(out, f"blocks.{b}.layers.{i}", dil)], dim=1)
                out = torch.relu(self.bn(out, f"transits.{b}.nonlinear.0", 1))
                out = self.conv1d(out, f"transits.{b}.linear")
            out = torch.relu(self.bn(out, "out_nonlinear.0", 1))
            mean = out.mean(dim=-1)
            std = torch.sqrt(out.var(dim=-1, unbiased=False) + 1e-5)
            st = torch.cat([mean, std], dim=-1)[:, :, None]
            emb = self.bn(self.conv1d(st, "dense.linear"), "dense.nonlinear.0", 1, affine=False)
            return emb[:, :, 0].numpy().astype(f32)

    def inference(self, audio16k: np.ndarray) -> np.ndarray:
        """CAMPPlus.inference (:788-818) for one clip: fbank -> subtract the per-bin time mean -> forward.  [n] -> [1, 192]."""
        fb = kaldi_fbank(audio16k)
        fb = fb - fb.mean(axis=0, keepdims=True, dtype=f32)
        return self.forward(fb[None])
