"""Kernel-level parity for the MFMA GEMM behind every Linear/Conv on the path, vs a plain fp32 torch matmul of the same
16-bit-rounded operands (floating-point kernel: tolerance = fp32 accumulation-order noise + one output rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, r, act, kind):
    from mlx_swift_audio_amd.synthetic import round_array
    y = torch.from_numpy(round_array(x, kind)) @ torch.from_numpy(round_array(w, kind)).t()
    if b is not None:
        y = y + torch.from_numpy(b)
    if act == "gelu":
        y = torch.nn.functional.gelu(y)
    if r is not None:
        y = y + torch.from_numpy(r)
    return y.numpy()


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
@pytest.mark.parametrize("M,N,K", [(1, 1, 64), (127, 129, 64), (300, 260, 192), (1500, 1280, 1280), (257, 3840, 320), (513, 520, 128), (3000, 1280, 384)])
def test_linear_shapes_and_tails(ctx, M, N, K, variant):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    got = ops.linear(ctx, x, w, b, r, act="gelu", dtype=m.BF16, out_f32=True, variant=variant)
    ref = _ref(x, w, b, r, "gelu", "bf16")
    np.testing.assert_allclose(got, ref, atol=2e-4, rtol=1e-4)       # fp32 out: only accumulation order differs
    got16 = ops.linear(ctx, x, w, b, None, act=None, dtype=m.F16, out_f32=False, variant=variant)
    ref16 = _ref(x, w, b, None, None, "f16")
    np.testing.assert_allclose(got16, ref16, atol=4e-3, rtol=2e-3)   # + one f16 rounding of the output


def test_linear_rejects_bad_k(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import ops
    with pytest.raises(m.MiaError):
        ops.linear(ctx, np.zeros((4, 48), np.float32), np.zeros((4, 48), np.float32))


def test_linear_8phase_is_deterministic(ctx):
    """Race screen for the LDS-DMA ring of the 8-phase tile: identical inputs must give bit-identical outputs on every launch
    (a DMA that lands after its reader, or a re-stage before the last read, shows up as run-to-run differences)."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import ops
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2100, 1280)).astype(np.float32)
    w = (rng.standard_normal((1300, 1280)) / 36.0).astype(np.float32)
    first = ops.linear(ctx, x, w, None, None, act=None, dtype=m.BF16, out_f32=True, variant=4)
    ref = ops.linear(ctx, x, w, None, None, act=None, dtype=m.BF16, out_f32=True, variant=1)
    np.testing.assert_allclose(first, ref, atol=2e-4, rtol=1e-4)
    for _ in range(12):
        again = ops.linear(ctx, x, w, None, None, act=None, dtype=m.BF16, out_f32=True, variant=4)
        assert np.array_equal(first, again)
