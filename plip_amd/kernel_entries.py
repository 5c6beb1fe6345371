"""Kernel-level entries of libplipmi.so (include/plipmi_test.h) for ``tests/`` and ``tools/`` -- NOT part of the product path.

``plip_amd.PLIP`` / ``PlipModel`` / ``Engine`` never call anything in here: these wrappers drive single kernels (the NT GEMM with every
epilogue, the attention kernels, the fused text q/k/v + attention kernel, the residual-plane re-coding) through the same C ABI so that the
GPU tests can compare them with fp64 references, plus the host mirrors of the split residual planes the tests check the kernels against.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .engine import _TORCH_DTYPE, _code, _ptr


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
            variant: int = -1, alpha: float = 1.0, out: Optional[torch.Tensor] = None,
            trace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Kernel-level entry (tests / micro-bench): epilogue(A[M,K] @ W[N,K]^T); a, w fp32 or bf16 CUDA tensors."""
    lib = _lib.load()
    assert a.is_cuda and w.is_cuda and a.dtype == w.dtype and a.is_contiguous() and w.is_contiguous()
    code = _code(a.dtype)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.zeros((M, N), dtype=a.dtype if epilogue in (0, 1) else torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.plipmi_gemm_nt_traced(code, epilogue, variant, M, N, K, _ptr(a), _ptr(w), _ptr(bias), float(alpha),
                                             _ptr(out), _ptr(trace),
                                             C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)),
                   "plipmi_gemm_nt")
    return out


def gemm_nt_ld(a: torch.Tensor, w: torch.Tensor, K: int, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
               variant: int = -1, alpha: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue(A[:, :K] @ W[:, :K]^T) on row-padded operands: a [M, lda >= K], w [N, ldw >= K] (tests)."""
    lib = _lib.load()
    assert a.is_cuda and w.is_cuda and a.dtype == w.dtype and a.is_contiguous() and w.is_contiguous()
    code = _code(a.dtype)
    M, N = a.shape[0], w.shape[0]
    if out is None:
        out = torch.zeros((M, N), dtype=a.dtype if epilogue in (0, 1) else torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.plipmi_gemm_nt_ld(code, epilogue, variant, M, N, K, _ptr(a), a.shape[1], _ptr(w), w.shape[1],
                                         _ptr(bias), float(alpha), _ptr(out),
                                         C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)), "plipmi_gemm_nt_ld")
    return out


def attention(qkv: torch.Tensor, B: int, S: int, H: int, causal: bool = False, key_mask: Optional[torch.Tensor] = None,
              impl: int = 0) -> torch.Tensor:
    """Kernel-level entry (tests): qkv [B*S, 3*H*64] (scale folded into q) -> [B*S, H*64]."""
    lib = _lib.load()
    assert qkv.is_cuda and qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * 64)
    code = _code(qkv.dtype)
    out = torch.empty((B * S, H * 64), dtype=qkv.dtype, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(lib.plipmi_attention(code, impl, _ptr(qkv), _ptr(out), B, S, H, int(causal), _ptr(key_mask),
                                        C.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)), "plipmi_attention")
    return out


def qkv_attention(a: torch.Tensor, w: torch.Tensor, c2: torch.Tensor, stats: torch.Tensor, B: int, S: int, H: int,
                  causal: bool = True, key_mask: Optional[torch.Tensor] = None, eps: float = 1e-5,
                  trace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Kernel-level entry (tests) of the fused LayerNorm-folded q/k/v projection + attention (plipmi_qkv_attention):
    a [B*S, 64H], w [3*64H, 64H] (16-bit), c2 [3*64H], stats [B*S, H, 2] -> attention output [B*S, 64H]."""
    lib = _lib.load()
    D = H * 64
    assert a.is_cuda and a.is_contiguous() and w.is_contiguous() and a.shape == (B * S, D) and w.shape == (3 * D, D) and w.dtype == a.dtype
    assert stats.shape == (B * S, H, 2) and stats.is_contiguous()
    out = torch.empty((B * S, D), dtype=a.dtype, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.plipmi_qkv_attention(_code(a.dtype), _ptr(a), _ptr(w), _ptr(c2), _ptr(stats), H, float(eps), _ptr(out),
                                            B, S, H, int(causal), _ptr(key_mask), _ptr(trace),
                                            C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)), "plipmi_qkv_attention")
    return out


def gemm_nt_ln(mode: int, a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, stats: Optional[torch.Tensor] = None,
               eps: float = 1e-5, variant: int = -1, out: Optional[torch.Tensor] = None):
    """Kernel-level entry (tests) for the LayerNorm-folded epilogues, see include/plipmi.h plipmi_gemm_nt_ln.
    mode 0/1 -> bf16 [M,N]; mode 2 -> (C fp32 updated in place, xb bf16 [M,N], st fp32 [M,N/64,2]); mode 3 -> the same
    update on the split residual stream: ``out`` = (hi 16-bit [M,N], lo uint8 [lo_plane_bytes(M, N)], the blocked 8-bit remainder
    plane), both updated in place; returns (hi, lo, st)."""
    lib = _lib.load()
    assert a.dtype in (torch.bfloat16, torch.float16) and w.dtype == a.dtype and a.is_contiguous() and w.is_contiguous()
    code, hdt = _code(a.dtype), a.dtype
    M, K = a.shape
    N = w.shape[0]
    stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
    with torch.cuda.device(a.device):
        if mode in (0, 1):
            if out is None:
                out = torch.empty((M, N), dtype=hdt, device=a.device)
            _lib.check(lib.plipmi_gemm_nt_ln(code, mode, variant, M, N, K, _ptr(a), _ptr(w), _ptr(bias), _ptr(stats),
                                             stats.shape[1], float(eps), _ptr(out), None, None, stream), "plipmi_gemm_nt_ln")
            return out
        st = torch.empty((M, N // 64, 2), dtype=torch.float32, device=a.device)
        if mode in (3, 4):           # 4: planes read in a's format, written in the other 16-bit type's (hi is then to be VIEWED as that type)
            hi, lo = out
            assert hi.dtype == hdt and lo.dtype == torch.uint8 and lo.numel() == lo_plane_bytes(M, N) and hi.is_contiguous() and lo.is_contiguous()
            _lib.check(lib.plipmi_gemm_nt_ln(code, mode, variant, M, N, K, _ptr(a), _ptr(w), _ptr(bias), None, 0, float(eps),
                                             _ptr(lo), _ptr(hi), _ptr(st), stream), "plipmi_gemm_nt_ln")
            return hi, lo, st
        xb = torch.empty((M, N), dtype=hdt, device=a.device)
        _lib.check(lib.plipmi_gemm_nt_ln(code, 2, variant, M, N, K, _ptr(a), _ptr(w), _ptr(bias), None, 0, float(eps),
                                         _ptr(out), _ptr(xb), _ptr(st), stream), "plipmi_gemm_nt_ln")
        return out, xb, st


def _pow2(k: torch.Tensor) -> torch.Tensor:
    """2^k as float64, built from the bit pattern (torch.pow / torch.ldexp are not exact on the GPU)"""
    return ((k.to(torch.int64) + 1023) << 52).view(torch.float64)


def lo_plane_index(M: int, N: int, device=None) -> torch.Tensor:
    """Byte offset of element (m, n) in the blocked lo plane (csrc/common.h lo_plane_off): blocks of 16 rows x 8 columns = 128 B,
    [row & 7][row >> 3 & 1][column & 7] inside a block, blocks column-major inside a 16-row band.  int64 [M, N]."""
    m = torch.arange(M, dtype=torch.int64, device=device)[:, None]
    n = torch.arange(N, dtype=torch.int64, device=device)[None, :]
    return (m >> 4) * 16 * N + (n >> 3) * 128 + (m & 7) * 16 + ((m >> 3) & 1) * 8 + (n & 7)


def lo_plane_bytes(M: int, N: int) -> int:
    return (M + 15) // 16 * 16 * N


def lo_plane_values(lo: torch.Tensor, M: int, N: int) -> torch.Tensor:
    """The remainders of rows 0 .. M-1 as int8 [M, N] (a band's rows past M are padding: kernels may leave anything there)."""
    return lo[lo_plane_index(M, N, lo.device).reshape(-1)].reshape(M, N).view(torch.int8)


def split_planes(x: torch.Tensor, dtype=torch.bfloat16):
    """fp32 [M, N] -> the engine's two-plane residual form (csrc/common.h split_f32<H>), on the host in torch integer / float64
    arithmetic, bit for bit what the kernels write: ``hi`` [M, N] in the operand type, ``lo`` = uint8 [lo_plane_bytes(M, N)] in the
    blocked layout (padding bytes zero).
    bf16: hi = nearest bf16 (ties away from zero); r = bits(x) - (hi << 16) in [-32768, 32767]; lo = min((r + 128) >> 8, 127).
    f16:  hi = nearest f16 (ties to even, saturating); lo = clamp(rint((x - hi) * 2^(18 - E(hi))), -128, 127), E >= -14."""
    assert x.dim() == 2
    M, N = x.shape
    if dtype == torch.bfloat16:
        u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        t = (u + 0x8000) & 0xFFFFFFFF
        hi = (t >> 16).to(torch.int32)
        r = (u - (t & 0xFFFF0000))                                   # in [-32768, 32767]
        lo = ((r + 128) >> 8).clamp(max=127)
        hi16 = torch.where(hi >= 32768, hi - 65536, hi).to(torch.int16).view(torch.bfloat16)
    else:
        assert dtype == torch.float16
        xc = x.contiguous().float()
        hi16 = xc.clamp(-65504.0, 65504.0).to(torch.float16)
        hf = hi16.float()
        eb = ((hf.view(torch.int32) >> 23) & 0xFF).clamp(min=113)
        lo = torch.round((xc.double() - hf.double()) * _pow2(145 - eb)).clamp(-128, 127).to(torch.int64)   # exact power-of-two scaling, ties to even
    plane = torch.zeros(lo_plane_bytes(M, N), dtype=torch.uint8, device=x.device)
    plane[lo_plane_index(M, N, x.device).reshape(-1)] = (lo & 0xFF).to(torch.uint8).reshape(-1)
    return hi16, plane


def join_planes(hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    """The fp32 value the two planes stand for (csrc/common.h join_f32<H>), bit for bit what the kernels read."""
    M, N = hi.shape
    l8 = lo[lo_plane_index(M, N, hi.device).reshape(-1)].reshape(M, N).to(torch.int64)
    l8 = torch.where(l8 >= 128, l8 - 256, l8)
    if hi.dtype == torch.float16:
        hf = hi.float()
        eb = ((hf.view(torch.int32) >> 23) & 0xFF).clamp(min=113)
        return (hf.double() + l8.double() * _pow2(eb - 145)).float()
    h = hi.view(torch.int16).to(torch.int64) & 0xFFFF
    u = ((h << 16) + (l8 << 8)) & 0xFFFFFFFF
    u = torch.where(u >= 2 ** 31, u - 2 ** 32, u)
    return u.to(torch.int32).view(torch.float32)


def recode_planes(hi: torch.Tensor, lo: torch.Tensor, to_dtype) -> tuple:
    """The residual planes re-coded in place for the other 16-bit operand type (plipmi_recode_planes): returns (hi, lo) views."""
    lib = _lib.load()
    frm = _code(hi.dtype)
    to = _code(to_dtype)
    M, N = hi.shape
    assert hi.is_cuda and lo.is_cuda and hi.is_contiguous() and lo.is_contiguous() and lo.dtype == torch.uint8 and lo.numel() == lo_plane_bytes(M, N)
    with torch.cuda.device(hi.device):
        _lib.check(lib.plipmi_recode_planes(_ptr(hi), _ptr(lo), M, N, frm, to,
                                            C.c_void_p(torch.cuda.current_stream(hi.device).cuda_stream)), "plipmi_recode_planes")
    return hi.view(_TORCH_DTYPE[to]), lo


def gemm_variant_built(dtype, variant: int) -> bool:
    return bool(_lib.load().plipmi_gemm_variant_built(_code(dtype), int(variant)))


def gemm_variants():
    lib = _lib.load()
    names, i = [], 0
    while True:
        n = lib.plipmi_gemm_variant_name(i)
        if not n:
            return names
        names.append(n.decode())
        i += 1
