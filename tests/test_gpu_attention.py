"""Kernel-level parity of the attention kernels (plipmi_attention) against an fp64 softmax(QK^T)V
of the same operands: exact-fp32 VALU kernel and bf16 MFMA kernel, vision (S=50, dense) and
text (S=77, causal + padding mask) shapes plus edge lengths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, B, S, H, causal, mask):
    D = H * 64
    x = qkv.double().reshape(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))      # [B,H,S,64]
    s = q @ k.transpose(-1, -2)                                         # scale already folded into q
    neg = float("-inf")
    if causal:
        keep = torch.tril(torch.ones(S, S, dtype=torch.bool, device=qkv.device))
        s = s.masked_fill(~keep, neg)
    if mask is not None:
        s = s.masked_fill(~mask.bool()[:, None, None, :], neg)
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, D)


CASES = [(3, 50, 12, False, False), (4, 77, 8, True, True), (2, 77, 8, True, False), (2, 1, 2, False, False),
         (2, 33, 2, True, True), (1, 128, 2, False, False), (2, 97, 2, True, True), (1, 64, 3, True, False),
         # S > 128: chunked online-softmax MFMA kernel (ViT-B/16 197, ViT-L/14 257, ViT-L/14@336 577 tokens)
         (2, 129, 2, False, False), (2, 197, 12, False, False), (2, 257, 16, False, True), (1, 577, 16, False, False),
         (2, 300, 2, True, True), (1, 256, 1, True, False), (1, 1000, 1, False, True)]


@pytest.mark.parametrize("B,S,H,causal,use_mask", CASES)
@pytest.mark.parametrize("mode", ["valu_f32", "valu_bf16", "mfma_bf16", "valu_f16", "mfma_f16"])
def test_attention_kernels(B, S, H, causal, use_mask, mode):
    from plip_amd.kernel_entries import attention
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 * S + H)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g)
    qkv[:, : H * 64] *= 0.125 * 3.0          # q pre-scaled; x3 sharpens the softmax
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode.split("_")[1]]
    qkv = qkv.to(dev).to(dtype)
    mask = None
    if use_mask:
        lens = torch.randint(1, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    out = attention(qkv, B, S, H, causal, mask, impl=1 if mode.startswith("mfma") else 0)
    torch.cuda.synchronize()
    ref = _ref(qkv, B, S, H, causal, mask)
    err = (out.double() - ref).abs().max().item()
    # fp32 kernel: roundoff; 16-bit I/O: output rounding 2^-9 (bf16) / 2^-12 (f16) * |o| (|o| <~ 4) and, for MFMA, P in that type
    tol = {"valu_f32": 2e-5, "valu_bf16": 2e-2, "mfma_bf16": 3e-2, "valu_f16": 2.5e-3, "mfma_f16": 4e-3}[mode]
    assert torch.isfinite(out).all()
    assert err < tol, f"{mode} B{B} S{S} H{H} causal={causal} mask={use_mask}: max err {err:.3e}"


FUSED = [(256, 77, 8, True, True), (5, 77, 8, True, False), (7, 65, 2, True, True), (9, 80, 4, True, True),
         (6, 77, 12, True, True), (4, 72, 2, False, True), (1, 77, 8, True, False)]


@pytest.mark.parametrize("B,S,H,causal,use_mask", FUSED)
@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_fused_qkv_attention_matches_the_two_kernels(B, S, H, causal, use_mask, hdt):
    """csrc/qkv_attention.hip -- the text tower's LayerNorm-folded q/k/v GEMM with the attention in its epilogue -- against
    the two kernels it replaces (plipmi_gemm_nt_ln mode 0 on the engine's own tile + plipmi_attention impl 1): the SAME
    BITS, for whole and partial caption groups, with and without the tokenizer mask; and against fp64."""
    from plip_amd.kernel_entries import attention, gemm_nt_ln, qkv_attention
    dev = torch.device("cuda:0")
    D = H * 64
    g = torch.Generator().manual_seed(1000 * S + 10 * H + B)
    x = torch.randn(B * S, D, generator=g) * 2.0 + 0.3                       # the LayerNorm input rows
    a = x.to(dev).to(hdt)
    xs = x.to(dev).reshape(B * S, H, 64)
    st = torch.stack((xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous()
    w = torch.randn(3 * D, D, generator=g) / D ** 0.5
    w = w - w.mean(-1, keepdim=True)                                         # folded weights have centred rows
    w[:D] *= 0.125 * 3.0                                                     # q pre-scaled; x3 sharpens the softmax
    w = w.to(dev).to(hdt)
    c2 = (torch.randn(3 * D, generator=g) * 0.2).to(dev)
    mask = None
    if use_mask:
        lens = torch.randint(1, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    qkv = gemm_nt_ln(0, a, w, c2, st, eps=1e-5, variant=-1)
    want = attention(qkv, B, S, H, causal, mask, impl=1)
    got = qkv_attention(a, w, c2, st, B, S, H, causal, mask, eps=1e-5)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), f"max |diff| {(got.float() - want.float()).abs().max().item():.3e}"
    ref = _ref(qkv, B, S, H, causal, mask)
    assert (got.double() - ref).abs().max().item() < (3e-2 if hdt == torch.bfloat16 else 4e-3)


def test_engine_runs_the_fused_text_kernel_and_the_two_kernels_to_the_same_bits(engines):
    """The bf16 engine's text tower (77 tokens) takes the fused kernel; the test hook that splits it back into GEMM + attention
    must not move a single bit of text_embeds, nor of the hidden states."""
    from plip_amd import _lib, weights as W
    lib = _lib.load()
    for dtype in ("bf16", "f16"):
        model, cfg, sd, *_ = engines("vitb32_b4", dtype, 256)
        ids, mask = W.synthetic_ids(cfg, 100, seed=77)
        ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
        try:
            lib.plipmi_test_fused_qkv_attention(0)
            two = model.get_text_features(input_ids=ids_t, attention_mask=mask_t)
            h_two = model.engine.hidden("text", 3, ids_t)
            lib.plipmi_test_fused_qkv_attention(1)
            one = model.get_text_features(input_ids=ids_t, attention_mask=mask_t)
            h_one = model.engine.hidden("text", 3, ids_t)
            lib.plipmi_test_fused_qkv_attention(2)                       # fused at a batch the product rule leaves to the two kernels
            small = model.get_text_features(input_ids=ids_t[:6], attention_mask=mask_t[:6])
        finally:
            lib.plipmi_test_reset_hooks()
        assert torch.equal(one, two) and torch.equal(h_one, h_two) and torch.equal(small, two[:6])

        def kernels(n):
            rows = []
            with model.engine.profile(rows):
                model.get_text_features(input_ids=ids_t[:n], attention_mask=mask_t[:n])
            return {r["name"].split("|")[0] for r in rows}
        big, few = kernels(100), kernels(40)
        # 100 captions x 8 heads = 200 workgroups: fused; 40 captions = 80 workgroups: the two kernels are the faster form there
        assert "qkv_attention" in big and not any(n.startswith("attention") for n in big), big
        assert "qkv_attention" not in few and any(n.startswith("attention") for n in few), few


def test_attention_argument_checks():
    from plip_amd._lib import PlipmiError
    from plip_amd.kernel_entries import attention
    qkv = torch.zeros(200, 3 * 64, device="cuda:0", dtype=torch.float32)
    with pytest.raises(PlipmiError):
        attention(qkv, 1, 200, 1, impl=1)            # the MFMA kernels take the 16-bit types only
    out = attention(qkv, 1, 200, 1, impl=0)          # the exact kernel takes any S <= 1024
    assert out.shape == (200, 64)
    long = torch.zeros(2000, 3 * 64, device="cuda:0", dtype=torch.bfloat16)
    with pytest.raises(PlipmiError):
        attention(long, 1, 2000, 1, impl=0)
    assert attention(long, 1, 2000, 1, impl=1).shape == (2000, 64)   # the chunked MFMA kernel has no length limit


def test_fully_masked_tail_chunks_do_not_disturb_the_running_softmax():
    """Keys 0..9 valid, everything after (two whole chunks) masked: the online softmax must equal the 10-key one."""
    from plip_amd.kernel_entries import attention
    g = torch.Generator().manual_seed(9)
    S, H = 300, 2
    qkv = torch.randn(S, 3 * H * 64, generator=g).to("cuda:0").to(torch.bfloat16)
    mask = (torch.arange(S)[None, :] < 10).long().to("cuda:0")
    out = attention(qkv, 1, S, H, False, mask, impl=1)
    ref = _ref(qkv, 1, S, H, False, mask)
    assert (out.double() - ref).abs().max().item() < 3e-2
    # and the mirrored case: only the LAST chunk has valid keys (first chunks contribute nothing, m stays -inf)
    mask2 = (torch.arange(S)[None, :] >= 290).long().to("cuda:0")
    out2 = attention(qkv, 1, S, H, False, mask2, impl=1)
    assert torch.isfinite(out2).all()
    assert (out2.double() - _ref(qkv, 1, S, H, False, mask2)).abs().max().item() < 3e-2


@pytest.mark.parametrize("S,H,causal", [(77, 2, True), (50, 2, False), (300, 2, False)])
def test_masked_key_rows_contribute_nothing_and_a_non_finite_one_poisons_like_hfs_additive_mask(S, H, causal):
    """The MFMA kernels mask ADDITIVELY (score + (0 | -inf): ADVICE r3 -- a select was miscompiled by hipcc 7.2's v_bitop3
    folding), which is also what HF does (attention_mask is added to the scores, modeling_clip.py:259-277).  Pinned here:
    (a) whatever FINITE values sit in masked key / value rows -- padding rows hold real embeddings, never zeros -- cannot
    reach any query; (b) a NaN in a masked KEY row does reach the queries that would have multiplied it (NaN + -inf = NaN
    -> the row maximum), exactly as in HF; it never appears in the engine, whose masked rows are finite activations."""
    from plip_amd.kernel_entries import attention
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(S)
    B, n_valid = 2, S // 2
    qkv = torch.randn(B * S, 3 * H * 64, generator=g)
    qkv[:, : H * 64] *= 0.125
    mask = (torch.arange(S)[None, :] < n_valid).long().repeat(B, 1).to(dev)
    base = qkv.to(dev).to(torch.bfloat16)
    want = attention(base, B, S, H, causal, mask, impl=1)
    loud = base.clone().view(B, S, 3, H * 64)
    loud[:, n_valid:, 1:] = 3.0e4                                   # masked K and V rows: huge but finite
    got = attention(loud.view(B * S, -1), B, S, H, causal, mask, impl=1)
    torch.cuda.synchronize()
    live = torch.zeros(B, S, dtype=torch.bool)
    live[:, :n_valid] = True                                        # (rows of masked QUERIES are don't-cares downstream)
    assert torch.equal(got.view(B, S, -1)[live.to(dev)], want.view(B, S, -1)[live.to(dev)])
    bad = base.clone().view(B, S, 3, H * 64)
    bad[0, S - 1, 1, :64] = float("nan")                            # one masked key row of sample 0, head 0
    out = attention(bad.view(B * S, -1), B, S, H, causal, mask, impl=1).view(B, S, H, 64)
    torch.cuda.synchronize()
    assert torch.isfinite(out[1]).all() and torch.isfinite(out[0, :, 1]).all()          # other sample / other head untouched
    if not causal:
        assert torch.isnan(out[0, :n_valid, 0]).all()               # additive mask: NaN + -inf stays NaN (as in HF)
