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
         (2, 33, 2, True, True), (1, 128, 2, False, False), (2, 97, 2, True, True), (1, 64, 3, True, False)]


@pytest.mark.parametrize("B,S,H,causal,use_mask", CASES)
@pytest.mark.parametrize("mode", ["valu_f32", "valu_bf16", "mfma_bf16"])
def test_attention_kernels(B, S, H, causal, use_mask, mode):
    from plip_amd.engine import attention
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 * S + H)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g)
    qkv[:, : H * 64] *= 0.125 * 3.0          # q pre-scaled; x3 sharpens the softmax
    dtype = torch.float32 if mode == "valu_f32" else torch.bfloat16
    qkv = qkv.to(dev).to(dtype)
    mask = None
    if use_mask:
        lens = torch.randint(1, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    out = attention(qkv, B, S, H, causal, mask, impl=1 if mode == "mfma_bf16" else 0)
    torch.cuda.synchronize()
    ref = _ref(qkv, B, S, H, causal, mask)
    err = (out.double() - ref).abs().max().item()
    # fp32 kernel: roundoff; bf16 I/O: output rounding 2^-9 * |o| (|o| <~ 4) and, for MFMA, bf16 P
    tol = {"valu_f32": 2e-5, "valu_bf16": 2e-2, "mfma_bf16": 3e-2}[mode]
    assert torch.isfinite(out).all()
    assert err < tol, f"{mode} B{B} S{S} H{H} causal={causal} mask={use_mask}: max err {err:.3e}"


def test_mfma_attention_rejects_long_sequences():
    from plip_amd._lib import PlipmiError
    from plip_amd.engine import attention
    qkv = torch.zeros(200, 3 * 64, device="cuda:0", dtype=torch.bfloat16)
    with pytest.raises(PlipmiError):
        attention(qkv, 1, 200, 1, impl=1)
    out = attention(qkv, 1, 200, 1, impl=0)          # the exact kernel takes any S <= 1024
    assert out.shape == (200, 64)
