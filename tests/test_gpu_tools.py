"""The measurement tools are code too (a mis-scaled probe steered half a round): run the in-kernel timeline sections
of tools/gpu_diag.py on small problems and check that what they print is self-consistent -- phases add up to lifetimes, the
tile count is the grid, FLOP rates are computed from the shape that was launched."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_diag.py"), *args], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _p50_cycles(out, label):
    m = re.search(re.escape(label) + r".*?p50\s+(\d+)", out)
    assert m, (label, out)
    return int(m.group(1))


@pytest.mark.parametrize("variant,bm,bn", [(6, 160, 256), (3, 320, 256), (2, 256, 256)])
def test_gemm_timeline_is_self_consistent(variant, bm, bn):
    M, N, K = 1600, 512, 512
    out = _run("gemmtrace", str(variant), str(M), str(N), str(K), "2")
    tiles = -(-M // bm) * (N // bn)
    assert f"{tiles} workgroups" in out and f"{M}x{N}x{K}" in out
    parts = sum(_p50_cycles(out, lab) for lab in ("prologue     :", "main loop    :", "epilogue     :"))
    life = _p50_cycles(out, "lifetime     :")
    assert abs(parts - life) < 0.15 * life, (parts, life)            # medians of the three phases ~ the median lifetime
    kt = int(re.search(r"cycles per k-tile, (\d+) tiles", out).group(1))
    assert kt == K // 64 - 1                                          # the loop's K tiles (the last one runs behind it)
    ev = re.search(r"kernel\s+([\d.]+) us by events \((\d+) TFLOP/s\)", out)
    assert abs(float(ev.group(2)) - 2.0 * M * N * K / float(ev.group(1)) / 1e6) <= 1.0   # the rate is THIS shape's FLOPs / that time


def test_fused_text_kernel_timeline_is_self_consistent():
    out = _run("qkvattn", "40", "77", "2")
    assert "20 workgroups" in out                                     # ceil(40 / 4) caption groups x 2 heads
    parts = sum(_p50_cycles(out, lab) for lab in ("first tile landed):", "K loop (2 tiles):", "q/k/v -> LDS images :", "attention + stores  :"))
    life = _p50_cycles(out, "lifetime            :")
    assert abs(parts - life) < 0.15 * life, (parts, life)
    m = re.search(r"warm .*?attention_mfma\s+([\d.]+) us\s+fused qkv_attention\s+([\d.]+) us", out)
    assert m and 1.0 < float(m.group(1)) < 1e4 and 1.0 < float(m.group(2)) < 1e4
