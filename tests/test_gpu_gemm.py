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


# Tile variants (csrc/gemm_inst.h): 0 / 1 = 128x128 (64-bit addresses / buffer LDS-DMA), 2 = 256x256, 3 = 320x256,
# 4 = 192x256, 5 = 160x256 (uneven wave rows: 3 + 2 row blocks), 6 = 160x256 on a ring of three LDS stages,
# (16-bit: wave rows of five 16-row MFMA tiles, half slabs in the epilogues); -2 = the naive checker kernel.
# (Round 5's tile 7 -- 160x128, two workgroups per CU -- was measured, lost, and is no longer built: profiles/r05_duo_tile.txt.)
HALF = {"bf16": torch.bfloat16, "f16": torch.float16}
NVAR = 7


def _built(dtype):
    from plip_amd.kernel_entries import gemm_variant_built, gemm_variants
    return [v for v in list(range(len(gemm_variants()))) + [-2] if gemm_variant_built(dtype, v)]


def test_every_listed_variant_is_built():
    from plip_amd.kernel_entries import gemm_variants
    assert len(gemm_variants()) == NVAR
    for dt in (torch.float32, *HALF.values()):
        assert _built(dt) == list(range(NVAR)) + [-2]


def _half_tol(y, ref):
    # 16-bit outputs: one RNE rounding (bf16 2^-9, f16 2^-12 relative); fp32 outputs: fp32 accumulation noise only
    if y.dtype == torch.float32:
        return 2e-4
    return (4e-3 if y.dtype == torch.bfloat16 else 6e-4) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("variant", list(range(NVAR)) + [-2])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_variants(dtype, variant, epi):
    from plip_amd.kernel_entries import gemm_nt, gemm_variant_built
    assert gemm_variant_built(dtype, variant)
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
        tol = _half_tol(y, ref)
        assert err < tol, f"variant {variant} epi {epi} {M}x{N}x{K}: max err {err:.3e} (tol {tol:.1e})"


def test_gemm_matches_naive_checker_bitwise_fp32():
    """fp32 MFMA is an fmaf chain: tiled and naive kernels only differ in summation order."""
    from plip_amd.kernel_entries import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    a = torch.randn(200, 256, generator=g).to(dev)
    w = (torch.randn(384, 256, generator=g) / 16).to(dev)
    b = torch.randn(384, generator=g).to(dev)
    y1 = gemm_nt(a, w, b, epilogue=0, variant=0)
    y0 = gemm_nt(a, w, b, epilogue=0, variant=1)
    yn = gemm_nt(a, w, b, epilogue=0, variant=-2)
    assert torch.equal(y0, y1)                      # same tile shape: 64-bit global LDS-DMA vs buffer LDS-DMA + fragment pipeline
    assert (y1 - yn).abs().max().item() < 1e-5


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_every_tile_sums_k_in_the_same_order(epi, hdt):
    """A row's embedding must not depend on the batch it arrives in (PLIP.coalesce, the embedding caches): small problems run
    on the 128x128 tile (v_mfma 32x32x16), large ones on the 16x16x32 tiles (2, 3, 6) -- so every tile has to produce the
    SAME bits from the same operands, i.e. the two MFMA shapes must sum K in the same order."""
    from plip_amd.kernel_entries import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77 + epi)
    for (M, N, K) in [(1300, 768, 768), (2100, 512, 2048), (1300, 1536, 512)]:
        a = torch.randn(M, K, generator=g).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(hdt)
        bias = torch.randn(N, generator=g).to(dev)
        c0 = torch.randn(M, N, generator=g).to(dev)
        run = lambda v: gemm_nt(a, w, bias, epilogue=epi, variant=v, alpha=0.37, out=c0.clone() if epi == 2 else None)
        y0 = run(0)
        for v in (1, 2, 3, 4, 5, 6):
            assert torch.equal(run(v), y0), f"tile {v} differs from the 128x128 tile: epi {epi} {M}x{N}x{K} {hdt}"


def test_gemm_rejects_bad_shapes():
    from plip_amd._lib import PlipmiError
    from plip_amd.kernel_entries import gemm_nt
    dev = torch.device("cuda:0")
    a = torch.zeros(8, 48, device=dev)
    w = torch.zeros(128, 48, device=dev)
    with pytest.raises(PlipmiError):
        gemm_nt(a, w, torch.zeros(128, device=dev), variant=1)      # K % 32 != 0


def test_operands_of_four_gib_take_the_64bit_address_kernels():
    """The buffer-DMA kernels carry 32-bit byte offsets; gemm_launch must route a 4 GiB A operand to their
    global-address twins -- rows beyond the 4 GiB mark have to come out right."""
    from plip_amd.kernel_entries import gemm_nt
    dev = torch.device("cuda:0")
    M, N, K = (1 << 20) + 37, 256, 2048                       # A = 4 GiB + 148 KiB of bf16
    a = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(5)
    a[:4096].normal_(generator=g)
    a[4096:-4096] = 0.0
    a[-4096:].normal_(generator=g)
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    for variant in (2, -1):
        y = gemm_nt(a, w, bias, epilogue=0, variant=variant)
        torch.cuda.synchronize()
        for sl in (slice(0, 300), slice(M - 300, M)):
            ref = a[sl].double() @ w.double().T + bias.double()
            assert (y[sl].double() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())
        assert y[5000:6000].float().sub(bias).abs().max().item() < 2e-2      # zero rows -> bias only
    del a, y
    torch.cuda.empty_cache()


# ---- LayerNorm folded into the GEMMs (gemm.h EPI_BIAS_LN / EPI_QGELU_LN / EPI_RESID_EMIT) ---------------------------
def _slice_stats(x):
    """per-row partials over 64-column slices: {sum, centred M2} -- what the producers emit"""
    M, D = x.shape
    xs = x.double().reshape(M, D // 64, 64)
    s = xs.sum(-1)
    m2 = ((xs - s[..., None] / 64) ** 2).sum(-1)
    return torch.stack((s, m2), dim=-1).float().contiguous()


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("variant", list(range(NVAR)) + [-1, -3])
@pytest.mark.parametrize("mode", [0, 1])
def test_layernorm_folded_consumer_epilogue(variant, mode, hdt):
    """y = [quickgelu](Linear(LayerNorm(x))) computed as rstd * (bf16(x) @ W'^T) + c2 with W' = bf16(W * g, rows centred)
    -- the centring of W' is what subtracts the row mean -- and rstd recombined from the 64-column partials (Chan),
    against an fp64 product of the same rounded operands and against the textbook LayerNorm -> Linear.
    Rows carry a common offset and one outlier channel, so a naive E[x^2] - mean^2 would lose digits."""
    from plip_amd.kernel_entries import gemm_nt_ln
    dev = torch.device("cuda:0")
    g0 = torch.Generator().manual_seed(100 + variant + 10 * mode)
    for (M, N, D) in [(1, 256, 128), (77, 512, 512), (300, 768, 768), (1200, 256, 1024)]:
        if variant == -3 and D % 256:
            continue                            # the split-K small-M kernel needs K % 256 == 0
        x = torch.randn(M, D, generator=g0) + 1.5
        x[:, 5] += 60.0
        W = torch.randn(N, D, generator=g0) / D ** 0.5
        bias = torch.randn(N, generator=g0) * 0.1
        gain = torch.exp(torch.empty(D).uniform_(-2.3, 2.3, generator=g0))
        beta = torch.randn(D, generator=g0)
        xb = x.to(dev).to(hdt)
        Wg = W * gain[None, :]
        Wf = (Wg - Wg.mean(1, keepdim=True)).to(dev).to(hdt)
        c2 = (W.double() @ beta.double() + bias.double()).float().to(dev)
        st = _slice_stats(x.to(dev))
        y = gemm_nt_ln(mode, xb, Wf, c2, st, eps=1e-5, variant=variant)
        torch.cuda.synchronize()
        # reference with the SAME rounded operands, statistics of the unrounded rows (as the engine has them)
        xd = x.to(dev).double()
        mu = xd.mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((xd - mu) ** 2).mean(1, keepdim=True) + 1e-5)
        ref = rstd * (xb.double() @ Wf.double().T) + c2.double()[None, :]
        if mode == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        scale = max(1.0, ref.abs().max().item())
        err = (y.double() - ref).abs().max().item()
        rnd = 6e-3 if hdt == torch.bfloat16 else 8e-4
        assert err < rnd * scale, f"variant {variant} mode {mode} {M}x{N}x{D}: err {err:.3e} (|ref| max {scale:.2f})"
        # and that IS LayerNorm -> Linear: compare with the textbook form in fp64 (bf16 operand rounding only)
        h = (xd - mu) * rstd * gain.to(dev).double() + beta.to(dev).double()
        book = h @ W.to(dev).double().T + bias.to(dev).double()
        if mode == 1:
            book = book * torch.sigmoid(1.702 * book)
        assert (y.double() - book).abs().max().item() < (4e-2 if hdt == torch.bfloat16 else 6e-3) * max(1.0, book.abs().max().item())


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("variant", list(range(NVAR)) + [-1, -3])
def test_layernorm_folded_producer_epilogue(variant, hdt):
    """x += A @ W^T + bias in place (fp32), plus the bf16 copy and the per-slice {sum, centred M2} of the updated rows,
    the latter against fp64 statistics of the kernel's OWN fp32 output (so the check is exact to fp32 round-off)."""
    from plip_amd.kernel_entries import gemm_nt_ln
    dev = torch.device("cuda:0")
    g0 = torch.Generator().manual_seed(200 + variant)
    for (M, N, K) in [(1, 256, 64), (50, 512, 512), (515, 768, 3072), (1300, 1024, 256)]:
        if variant == -3 and K % 256:
            continue
        a = torch.randn(M, K, generator=g0).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g0) / K ** 0.5).to(dev).to(hdt)
        bias = torch.randn(N, generator=g0).to(dev)
        x0 = (torch.randn(M, N, generator=g0) * 3.0 + 11.0).to(dev)
        x0[:, 7] -= 90.0
        x, xb, st = gemm_nt_ln(2, a, w, bias, variant=variant, out=x0.clone())
        torch.cuda.synchronize()
        ref = x0.double() + a.double() @ w.double().T + bias.double()
        assert (x.double() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
        assert torch.equal(xb, x.to(hdt))                                      # RNE of the very fp32 value stored
        want = _slice_stats(x)
        assert (st[..., 0] - want[..., 0]).abs().max().item() < 1e-3           # sums of 64 values around |x| ~ 10..100
        rel = ((st[..., 1] - want[..., 1]).abs() / want[..., 1].clamp(min=1e-3)).max().item()
        assert rel < 1e-4, rel


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("variant", list(range(NVAR)) + [-1])
def test_split_plane_residual_epilogue(variant, hdt):
    """The engine's residual update: the stream lives as two planes, hi = the value rounded to the operand type (the next GEMM's A
    operand), lo = an 8-bit remainder (csrc/common.h split_f32; rounds 2-5: 16 bits = the exact fp32 value).  The kernel must
    (a) read the planes back to the very value the format defines, (b) compute the update in fp32 exactly as the plain-array
    epilogue (mode 2) does from that value and store it as the format defines -- its planes are the HOST split of mode 2's fp32 result,
    bit for bit, hi being the correctly rounded operand --, and (c) emit the statistics of the updated rows."""
    from plip_amd.kernel_entries import gemm_nt_ln, join_planes, lo_plane_values, split_planes
    dev = torch.device("cuda:0")
    g0 = torch.Generator().manual_seed(300 + variant)
    rel = 2.0 ** -15 if hdt == torch.bfloat16 else 2.0 ** -18      # worst case of the format (the +128 -> +127 clamp corner)
    for (M, N, K) in [(1, 256, 64), (50, 512, 512), (515, 768, 3072), (1300, 1024, 256)]:
        a = torch.randn(M, K, generator=g0).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g0) / K ** 0.5).to(dev).to(hdt)
        bias = torch.randn(N, generator=g0).to(dev)
        x0 = (torch.randn(M, N, generator=g0) * 3.0 + 11.0).to(dev)
        x0[:, 7] -= 90.0
        if M > 1:                                                   # extreme values (row 0 is left out of the statistics check)
            x0[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -3.0e38] if hdt == torch.bfloat16 else [0.0, 3.0517578125e-05, 6.2e-5, -60000.0], device=dev)
        hi0, lo0 = split_planes(x0, hdt)
        x0q = join_planes(hi0, lo0)                                 # the stream value the planes stand for
        assert ((x0q - x0).abs() <= x0.abs() * rel + 2.0 ** -32).all()
        x_ref, xb_ref, st_ref = gemm_nt_ln(2, a, w, bias, variant=variant, out=x0q.clone())
        hi, lo, st = gemm_nt_ln(3, a, w, bias, variant=variant, out=(hi0.clone(), lo0.clone()))
        torch.cuda.synchronize()
        want_hi, want_lo = split_planes(x_ref, hdt)                 # same fp32 arithmetic, then the format's rounding of the remainder
        assert torch.equal(hi.view(torch.int16), want_hi.view(torch.int16)) and torch.equal(lo_plane_values(lo, M, N), lo_plane_values(want_lo, M, N))
        x = join_planes(hi, lo)
        assert ((x - x_ref).abs() <= x_ref.abs() * rel + 2.0 ** -32).all()
        r0 = 1 if M > 1 else 0
        want = _slice_stats(x_ref)[r0:]                             # the statistics describe the fp32 rows the epilogue computed
        assert (st[r0:, :, 0] - want[..., 0]).abs().max().item() < 1e-3
        assert ((st[r0:, :, 1] - want[..., 1]).abs() / want[..., 1].clamp(min=1e-3)).max().item() < 1e-4
        if hdt == torch.bfloat16:   # hi is the bf16 nearest to the fp32 result (ties away from zero; RNE differs only on exact ties)
            diff = hi.view(torch.int16).to(torch.int32) - xb_ref.view(torch.int16).to(torch.int32)
            ties = (x_ref.view(torch.int32) & 0xFFFF) == 0x8000
            assert (diff[~ties] == 0).all() and (diff.abs() <= 1).all()
        else:                       # f16: ties to even, i.e. exactly torch's conversion
            assert torch.equal(hi[r0:], xb_ref[r0:])
        ref = x0q.double() + a.double() @ w.double().T + bias.double()
        assert (x.double() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("variant", [2, 3, 6, -1])
def test_split_plane_epilogue_hands_the_stream_to_the_other_operand_type(variant, hdt):
    """Mode 4 = mode 3 writing the planes in the OTHER 16-bit type's split format: what the last f16 block of a mixed text tower
    (plipmi_config.text_f16_layers) does instead of a re-coding pass.  It splits the fp32 value the epilogue computed -- the planes
    are the host split (other type) of the plain-array result, bit for bit, with mode 3's statistics; a re-coding pass behind mode 3
    lands within one more rounding of the remainder of it."""
    from plip_amd.kernel_entries import gemm_nt_ln, join_planes, lo_plane_values, recode_planes, split_planes
    dev = torch.device("cuda:0")
    other = torch.float16 if hdt == torch.bfloat16 else torch.bfloat16
    g0 = torch.Generator().manual_seed(400 + variant)
    for (M, N, K) in [(50, 512, 512), (1300, 512, 2048)]:
        a = torch.randn(M, K, generator=g0).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g0) / K ** 0.5).to(dev).to(hdt)
        bias = torch.randn(N, generator=g0).to(dev)
        x0 = (torch.randn(M, N, generator=g0) * 3.0 + 1.0).to(dev)
        hi0, lo0 = split_planes(x0, hdt)
        x_ref, _, _ = gemm_nt_ln(2, a, w, bias, variant=variant, out=join_planes(hi0, lo0))
        hi3, lo3, st3 = gemm_nt_ln(3, a, w, bias, variant=variant, out=(hi0.clone(), lo0.clone()))
        hi4, lo4, st4 = gemm_nt_ln(4, a, w, bias, variant=variant, out=(hi0.clone(), lo0.clone()))
        hi4 = hi4.view(other)
        torch.cuda.synchronize()
        want_hi, want_lo = split_planes(x_ref, other)
        assert torch.equal(hi4.view(torch.int16), want_hi.view(torch.int16)) and torch.equal(lo_plane_values(lo4, M, N), lo_plane_values(want_lo, M, N)) and torch.equal(st4, st3)
        hi3r, lo3r = recode_planes(hi3, lo3, other)
        torch.cuda.synchronize()
        xr, x4 = join_planes(hi3r, lo3r), join_planes(hi4, lo4)
        assert ((xr - x4).abs() <= x_ref.abs() * 2.0 ** -14 + 2.0 ** -30).all()


@pytest.mark.parametrize("hdt", list(HALF.values()), ids=list(HALF))
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_small_m_split_k_gemm(epi, hdt):
    """gemm_skinny.hip (variant -3): 32 x 64 tile per workgroup, K split over its four waves with a k permutation shared by
    both operands, operands straight from L2 -- against fp64 and, bit for bit across batch sizes, against itself (a row's
    result must not depend on how many other rows the call carries: the pooled last block relies on it)."""
    from plip_amd.kernel_entries import gemm_nt
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31 + epi)
    for (M, N, K) in [(1, 64, 256), (8, 768, 768), (256, 768, 3072), (256, 2048, 512), (333, 512, 2048)]:
        a = torch.randn(M, K, generator=g).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(hdt)      # asymmetric operands
        bias = torch.randn(N, generator=g).to(dev)
        c0 = torch.randn(M, N, generator=g).to(dev)
        y = gemm_nt(a, w, bias, epilogue=epi, variant=-3, out=c0.clone() if epi == 2 else None)
        torch.cuda.synchronize()
        ref = _ref(a, w, bias, epi, 1.0, c0)
        assert (y.double() - ref).abs().max().item() < _half_tol(y, ref), (epi, M, N, K)
        if M >= 8:                                  # rows 3..7 alone give the same bits
            y2 = gemm_nt(a[3:8].contiguous(), w, bias, epilogue=epi, variant=-3, out=c0[3:8].clone() if epi == 2 else None)
            assert torch.equal(y2, y[3:8])
