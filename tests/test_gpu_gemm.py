"""Kernel-level parity of the MFMA NT GEMM (plipmi_gemm_nt) on the MI355X: every tile
variant x dtype x epilogue against an fp64 product of the same (already rounded) operands,
and against the naive one-thread-per-output checker kernel of the library."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1, 256, 64), (37, 256, 128), (128, 256, 192), (300, 512, 768), (515, 256, 3072),
          (5000, 768, 256)]       # 40 x 6 tiles of 128^2: more tiles than resident workgroups per XCD strip step


def _ref(a, w, bias, epi, alpha, c0):
    y = a.double() @ w.double().T
    if epi in (0, 1, 2):
        y = y + bias.double()
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    if epi == 2:
        y = y + c0.double()
    if epi == 3:
        y = alpha * y
    return y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, -2])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_variants(dtype, variant, epi):
    from plip_amd.engine import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1234 + 10 * epi + variant)
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, generator=g).to(dev).to(dtype)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dtype)      # asymmetric operands
        bias = torch.randn(N, generator=g).to(dev)
        c0 = torch.randn(M, N, generator=g).to(dev)
        out = c0.clone() if epi == 2 else None
        y = gemm_nt(a, w, bias, epilogue=epi, variant=variant, alpha=0.37, out=out)
        torch.cuda.synchronize()
        ref = _ref(a, w, bias, epi, 0.37, c0)
        err = (y.double() - ref).abs().max().item()
        # fp32 outputs: fp32 accumulation noise only; bf16 outputs: one RNE rounding (2^-9 relative)
        tol = 2e-4 if y.dtype == torch.float32 else 4e-3 * max(1.0, ref.abs().max().item())
        assert err < tol, f"variant {variant} epi {epi} {M}x{N}x{K}: max err {err:.3e} (tol {tol:.1e})"


def test_gemm_matches_naive_checker_bitwise_fp32():
    """fp32 MFMA is an fmaf chain: tiled and naive kernels only differ in summation order."""
    from plip_amd.engine import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    a = torch.randn(200, 256, generator=g).to(dev)
    w = (torch.randn(384, 256, generator=g) / 16).to(dev)
    b = torch.randn(384, generator=g).to(dev)
    y1 = gemm_nt(a, w, b, epilogue=0, variant=1)
    y0 = gemm_nt(a, w, b, epilogue=0, variant=0)
    yn = gemm_nt(a, w, b, epilogue=0, variant=-2)
    assert torch.equal(y0, y1)                      # same tile shape, LDS-DMA vs register staging
    assert (y1 - yn).abs().max().item() < 1e-5


def test_gemm_rejects_bad_shapes():
    from plip_amd._lib import PlipmiError
    from plip_amd.engine import gemm_nt
    dev = torch.device("cuda:0")
    a = torch.zeros(8, 48, device=dev)
    w = torch.zeros(128, 48, device=dev)
    with pytest.raises(PlipmiError):
        gemm_nt(a, w, torch.zeros(128, device=dev), variant=1)      # K % 32 != 0


def test_operands_of_four_gib_take_the_64bit_address_kernels():
    """The buffer-DMA kernels carry 32-bit byte offsets; gemm_launch must route a 4 GiB A operand to their
    global-address twins -- rows beyond the 4 GiB mark have to come out right."""
    from plip_amd.engine import gemm_nt
    dev = torch.device("cuda:0")
    M, N, K = (1 << 20) + 37, 256, 2048                       # A = 4 GiB + 148 KiB of bf16
    a = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(5)
    a[:4096].normal_(generator=g)
    a[4096:-4096] = 0.0
    a[-4096:].normal_(generator=g)
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    for variant in (35, -1):
        y = gemm_nt(a, w, bias, epilogue=0, variant=variant)
        torch.cuda.synchronize()
        for sl in (slice(0, 300), slice(M - 300, M)):
            ref = a[sl].double() @ w.double().T + bias.double()
            assert (y[sl].double() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())
        assert y[5000:6000].float().sub(bias).abs().max().item() < 2e-2      # zero rows -> bias only
    del a, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("epi", [0, 1])
def test_fp8_gemm_probe_is_exact_up_to_output_rounding(variant, epi):
    """EXPERIMENTAL fp8 (e4m3fn) operands through v_mfma_scale_f32_32x32x64_f8f6f4 (the configs[4] headroom probe):
    products of fp8 values are exact in fp32, so against an fp64 product of the SAME fp8 operands only fp32
    accumulation noise and the bf16 output rounding (half an ulp = 2^-9 relative) remain."""
    from plip_amd.engine import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77 + variant)
    for (M, N, K) in ((1, 256, 128), (300, 256, 128), (515, 512, 768), (1000, 768, 3072)):
        a = torch.randn(M, K, generator=g).to(dev).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, generator=g) / K ** 0.5 * 4).to(dev).to(torch.float8_e4m3fn)
        bias = torch.randn(N, generator=g).to(dev)
        y = gemm_nt(a, w, bias, epilogue=epi, variant=variant)
        torch.cuda.synchronize()
        ref = a.float().double() @ w.float().double().T + bias.double()
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        assert y.dtype == torch.bfloat16
        tol = 2.0 ** -8 * torch.clamp(ref.abs(), min=1.0) + 1e-3          # half a bf16 ulp, with slack for the epilogue
        assert bool(((y.double() - ref).abs() <= tol).all()), f"variant {variant} epi {epi} {M}x{N}x{K}"
