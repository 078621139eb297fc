"""Second opinions for the primitives the flow / S3 / codec / CAM++ restatements lean on (CPU, torch importable here).

The reference's arithmetic lives in MLX ops that are not in the tree: the oracles restate them with torch calls
wrapped for MLX's layouts (channels-last activations, weights [Cout, K, Cin/groups] / [Cout, KH, KW, Cin]).  Two kinds of check:
  * layout wrappers vs the ops' published DEFINITIONS written as explicit loops (y[t][o] = sum_{k,c} x[t*s + k*d - p][c] w[o][k][c]),
    including strides, dilations, groups / depthwise and the transposed convolution;
  * composite blocks vs an independent torch implementation: attention vs torch scaled_dot_product_attention (the d^-1/4 split
    scaling of S3 / Whisper, the 1/sqrt(d) of the flow; the conformer's positional term switched off), LayerNorm vs the
    mean / variance definition in float64, GELU / Mish / snake vs their closed forms.
Parity with the reference itself stays unpinned (no golden vectors exist); these pin the oracle to the documented op semantics."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import campplus as OCP
from oracle import codec as OC
from oracle import flow as OF
from oracle import s3tok as OS


def _conv1d_def(x_tc, w_okc, b, stride=1, pad=0, dil=1, groups=1):
    """MLX conv1d definition on [T, Cin] with weight [Cout, K, Cin/groups]: explicit loops, float64."""
    T, Cin = x_tc.shape
    Cout, K, Cg = w_okc.shape
    xp = np.zeros((T + 2 * pad, Cin), np.float64)
    xp[pad:pad + T] = x_tc
    T_out = (T + 2 * pad - dil * (K - 1) - 1) // stride + 1
    y = np.zeros((T_out, Cout), np.float64)
    og = Cout // groups
    for o in range(Cout):
        g = o // og
        for t in range(T_out):
            acc = 0.0
            for k in range(K):
                acc += float(np.dot(xp[t * stride + k * dil, g * Cg:(g + 1) * Cg], w_okc[o, k]))
            y[t, o] = acc + (0.0 if b is None else b[o])
    return y


def test_conv_wrappers_match_the_definition():
    rng = np.random.default_rng(0)
    # s3tok: channels-last [B, L, C], stride 2 pad 1 (the two subsampling convs) and the depthwise FSMN memory (groups = C, K = 31)
    x = rng.standard_normal((1, 23, 6)).astype(np.float32)
    w = rng.standard_normal((8, 3, 6)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    got = OS._conv1d_cl(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 2, 1)[0].numpy()
    np.testing.assert_allclose(got, _conv1d_def(x[0], w, b, stride=2, pad=1), atol=1e-5)
    wd = rng.standard_normal((6, 31, 1)).astype(np.float32)
    got = OS._conv1d_cl(torch.from_numpy(np.pad(x, ((0, 0), (15, 15), (0, 0)))), torch.from_numpy(wd), None, 1, 0, groups=6)[0].numpy()
    np.testing.assert_allclose(got, _conv1d_def(x[0], wd, None, pad=15, groups=6), atol=1e-5)
    # codec: channels-first wrapper, dilated depthwise k7 (SNAC residual unit) and plain k7
    xc = rng.standard_normal((1, 4, 30)).astype(np.float32)
    wdw = rng.standard_normal((4, 7, 1)).astype(np.float32)
    got = OC._conv1d_cf(torch.from_numpy(xc), torch.from_numpy(wdw), None, padding=9, dilation=3, groups=4)[0].numpy().T
    np.testing.assert_allclose(got, _conv1d_def(xc[0].T, wdw, None, pad=9, dil=3, groups=4), atol=1e-5)
    wk = rng.standard_normal((5, 7, 4)).astype(np.float32)
    got = OC._conv1d_cf(torch.from_numpy(xc), torch.from_numpy(wk), torch.from_numpy(b[:5]), padding=3)[0].numpy().T
    np.testing.assert_allclose(got, _conv1d_def(xc[0].T, wk, b[:5], pad=3), atol=1e-5)
    # flow: time-major [T, C], no padding, weight dict
    wf = {"c.weight": rng.standard_normal((7, 4, 5)).astype(np.float32), "c.bias": rng.standard_normal(7).astype(np.float32)}
    xt = rng.standard_normal((19, 5)).astype(np.float32)
    np.testing.assert_allclose(OF._conv(wf, "c", torch.from_numpy(xt)).numpy(), _conv1d_def(xt, wf["c.weight"], wf["c.bias"]), atol=1e-5)
    # CAM++: conv1d [B, C, T] with dilation, conv2d with MLX weight layout [Cout, KH, KW, Cin]
    ora = OCP.CAMPPlusOracle({"k.weight": rng.standard_normal((6, 3, 4)).astype(np.float32),
                              "q.weight": rng.standard_normal((3, 3, 3, 2)).astype(np.float32)})
    got = ora.conv1d(torch.from_numpy(xc), "k", stride=1, pad=2, dil=2)[0].numpy().T
    np.testing.assert_allclose(got, _conv1d_def(xc[0].T, ora.w["k.weight"], None, pad=2, dil=2), atol=1e-5)
    x2 = rng.standard_normal((1, 2, 9, 11)).astype(np.float32)          # [B, C, H, W]
    got = ora.conv2d(torch.from_numpy(x2), "q", (2, 1), 1)[0].numpy()
    w2 = ora.w["q.weight"]
    xp = np.pad(x2[0], ((0, 0), (1, 1), (1, 1))).astype(np.float64)
    Ho, Wo = (9 + 2 - 3) // 2 + 1, 11
    ref = np.zeros((3, Ho, Wo))
    for o in range(3):
        for i in range(Ho):
            for j in range(Wo):
                ref[o, i, j] = sum(xp[c, i * 2 + kh, j + kw] * w2[o, kh, kw, c] for c in range(2) for kh in range(3) for kw in range(3))
    np.testing.assert_allclose(got, ref, atol=1e-5)


def test_transposed_conv_wrapper_matches_torch_and_scatter():
    """MLX convTransposed1d with weight [Cout, K, Cin] == torch conv_transpose1d with weight [Cin, Cout, K] == scatter-add definition."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 3, 6)).astype(np.float32)
    w = rng.standard_normal((4, 8, 3)).astype(np.float32)
    for s, p in ((4, 2), (2, 1), (5, 3)):
        got = OC._convt1d_cf(torch.from_numpy(x), torch.from_numpy(w), None, s, p)[0].numpy()
        T_out = (6 - 1) * s - 2 * p + 8
        ref = np.zeros((4, T_out))
        for t in range(6):
            for k in range(8):
                j = t * s + k - p
                if 0 <= j < T_out:
                    ref[:, j] += w[:, k, :].astype(np.float64) @ x[0, :, t]
        np.testing.assert_allclose(got, ref, atol=1e-5)


def test_layernorm_and_activations_match_definitions():
    rng = np.random.default_rng(2)
    x = (3.0 * rng.standard_normal((5, 48)) + 1.5).astype(np.float32)
    w = {"n.weight": rng.standard_normal(48).astype(np.float32), "n.bias": rng.standard_normal(48).astype(np.float32)}
    for eps in (1e-5, 1e-12):
        x64 = x.astype(np.float64)
        ref = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + eps) * w["n.weight"] + w["n.bias"]
        np.testing.assert_allclose(OF._ln(w, "n", torch.from_numpy(x), eps).numpy(), ref, atol=2e-5)
    t = torch.linspace(-6, 6, 101)
    np.testing.assert_allclose(OF.mish(t).numpy(), (t.double() * torch.tanh(F.softplus(t.double()))).numpy(), atol=1e-6)
    np.testing.assert_allclose(F.gelu(t).numpy(), (0.5 * t.double() * (1 + torch.erf(t.double() / math.sqrt(2)))).numpy(), atol=1e-6)   # exact-erf GELU
    al = torch.tensor([0.5, 2.0]).view(1, 2, 1)
    xs = torch.randn(1, 2, 7)
    np.testing.assert_allclose(OC._snake(xs, al).numpy(), (xs + torch.sin(al * xs) ** 2 / (al + 1e-9)).numpy(), atol=1e-6)


def test_attention_blocks_match_sdpa():
    rng = np.random.default_rng(3)
    T, H, dk = 13, 2, 64
    D = H * dk

    def rnd(*s):
        return (rng.standard_normal(s) * 0.2).astype(np.float32)

    # flow conformer attention with the positional term switched off (zero linear_pos): (q + u) k^T / sqrt(dk) -> SDPA on q + u
    w = {f"a.linear_{n}.weight": rnd(D, D) for n in ("q", "k", "v", "out")}
    w.update({f"a.linear_{n}.bias": rnd(D) for n in ("q", "k", "v", "out")})
    w["a.linear_pos.weight"] = np.zeros((D, D), np.float32)
    w["a.pos_bias_u"], w["a.pos_bias_v"] = rnd(H, dk), rnd(H, dk)
    x = torch.from_numpy(rnd(T, D))
    got = OF.rel_attention(w, "a", x, OF.sinusoid_pe(T, D), H)
    q = (F.linear(x, torch.from_numpy(w["a.linear_q.weight"]), torch.from_numpy(w["a.linear_q.bias"])).reshape(T, H, dk) + torch.from_numpy(w["a.pos_bias_u"])).permute(1, 0, 2)
    k = F.linear(x, torch.from_numpy(w["a.linear_k.weight"]), torch.from_numpy(w["a.linear_k.bias"])).reshape(T, H, dk).permute(1, 0, 2)
    v = F.linear(x, torch.from_numpy(w["a.linear_v.weight"]), torch.from_numpy(w["a.linear_v.bias"])).reshape(T, H, dk).permute(1, 0, 2)
    o = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(T, D)
    ref = F.linear(o, torch.from_numpy(w["a.linear_out.weight"]), torch.from_numpy(w["a.linear_out.bias"]))
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)

    # estimator transformer block: attention without q/k/v bias at scale 64^-1/2, GELU feed-forward
    wt = {"t.norm1.weight": 1 + rnd(D), "t.norm1.bias": rnd(D), "t.norm3.weight": 1 + rnd(D), "t.norm3.bias": rnd(D),
          "t.attn.query_proj.weight": rnd(D, D), "t.attn.key_proj.weight": rnd(D, D), "t.attn.value_proj.weight": rnd(D, D),
          "t.attn.out_proj.weight": rnd(D, D), "t.attn.out_proj.bias": rnd(D),
          "t.ff.layers.0.weight": rnd(2 * D, D), "t.ff.layers.0.bias": rnd(2 * D), "t.ff.layers.1.weight": rnd(D, 2 * D), "t.ff.layers.1.bias": rnd(D)}
    got = OF.transformer(wt, "t", x, H)
    tt = lambda k: torch.from_numpy(wt[k])
    n = F.layer_norm(x, (D,), tt("t.norm1.weight"), tt("t.norm1.bias"), 1e-5)
    q, k, v = (F.linear(n, tt(f"t.attn.{nm}_proj.weight")).reshape(T, H, 64).permute(1, 0, 2)[None] for nm in ("query", "key", "value"))
    o = F.scaled_dot_product_attention(q, k, v)[0].permute(1, 0, 2).reshape(T, D)
    y = x + F.linear(o, tt("t.attn.out_proj.weight"), tt("t.attn.out_proj.bias"))
    n = F.layer_norm(y, (D,), tt("t.norm3.weight"), tt("t.norm3.bias"), 1e-5)
    ref = y + F.linear(F.gelu(F.linear(n, tt("t.ff.layers.0.weight"), tt("t.ff.layers.0.bias"))), tt("t.ff.layers.1.weight"), tt("t.ff.layers.1.bias"))
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=3e-5)


def test_s3_block_matches_an_independent_implementation():
    """One FSMN attention block of the S3 tokenizer (d^-1/4 scaling on q and k, RoPE before the head transpose, depthwise memory on V added
    after the output projection, additive -1e10 pad bias) rebuilt with torch SDPA + explicit RoPE; padded frames are masked."""
    from mlx_swift_audio_amd import synthetic as S
    cfg = S.S3Config(128, 128, 2, 1)
    w = S.s3_weights(cfg, 5)
    ora = OS.S3Oracle(cfg, w)
    rng = np.random.default_rng(4)
    B, T, D, H = 2, 20, 128, 2
    x = torch.from_numpy(rng.standard_normal((B, T, D)).astype(np.float32))
    lens = torch.tensor([20, 13])
    m = (torch.arange(T)[None, :] < lens[:, None]).float()
    got = ora._block("encoder.blocks.0", x, ((1.0 - m) * -1.0e10)[:, None, :], m[:, :, None])
    W = {k: torch.from_numpy(v) for k, v in w.items()}
    p = "encoder.blocks.0"
    h = F.layer_norm(x, (D,), W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"], 1e-5)
    q = F.linear(h, W[p + ".attn.query.weight"], W[p + ".attn.query.bias"]).reshape(B, T, H, 64)
    k = F.linear(h, W[p + ".attn.key.weight"]).reshape(B, T, H, 64)
    v = F.linear(h, W[p + ".attn.value.weight"], W[p + ".attn.value.bias"]).reshape(B, T, H, 64)
    # the port's table uses the exponent j / dim for j < dim / 2 (S3Tokenizer.swift:19-21), i.e. theta^-(j/64), NOT the usual 2j/dim
    inv = 1.0 / (10000.0 ** (torch.arange(32, dtype=torch.float32) / 64.0))
    ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]

    def rope(z):                                                                    # pairs (j, j + 32): [x_L cos - x_R sin, x_R cos + x_L sin]
        z1, z2 = z[..., :32], z[..., 32:]
        c, s = torch.cos(ang)[None, :, None, :], torch.sin(ang)[None, :, None, :]
        return torch.cat([z1 * c - z2 * s, z2 * c + z1 * s], dim=-1)

    mask = (m[:, None, None, :] > 0)                                                # keys of padded frames are excluded
    o = F.scaled_dot_product_attention(rope(q).transpose(1, 2), rope(k).transpose(1, 2), v.transpose(1, 2), attn_mask=mask)   # scale 1/sqrt(64) == d^-1/4 on each
    o = o.transpose(1, 2).reshape(B, T, D)
    vi = v.reshape(B, T, D) * m[:, :, None]
    mem = F.conv1d(F.pad(vi.transpose(1, 2), (15, 15)), W[p + ".attn.fsmn_block.weight"].permute(0, 2, 1), None, groups=D).transpose(1, 2) + vi
    y = x + F.linear(o, W[p + ".attn.out.weight"], W[p + ".attn.out.bias"]) + mem * m[:, :, None]
    h = F.layer_norm(y, (D,), W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"], 1e-5)
    ref = y + F.linear(F.gelu(F.linear(h, W[p + ".mlp.layers.0.weight"], W[p + ".mlp.layers.0.bias"])), W[p + ".mlp.layers.2.weight"], W[p + ".mlp.layers.2.bias"])
    for b in range(B):                                                              # valid frames only: padded rows are never read downstream
        np.testing.assert_allclose(got[b, :lens[b]].numpy(), ref[b, :lens[b]].numpy(), atol=5e-5)
