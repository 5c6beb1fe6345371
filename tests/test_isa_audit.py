"""Build-time ISA audit of the im2col-on-load patch GEMM (csrc/gemm.h ADDR 2).

``pix_load16`` issues an asynchronous ``buffer_load_dwordx4`` through inline asm with a plain "=v" output, so hipcc believes the
destination VGPRs are defined the moment the statement has issued, while the data is still in flight.  What keeps a later compiler
(or higher register pressure) from scheduling a ``v_mov`` / spill of those registers in front of the kernel's own ``s_waitcnt vmcnt``
is only the statement order and ``tie_regs5`` -- it happened once (DESIGN_HISTORY.md, round 5).  The run-time guard is the bit-identity
test on the GPU; this is the guard that runs wherever the library is BUILT: disassemble the gather kernels and assert that between
every register-destination ``buffer_load_dwordx4 / x3 ... offen`` (fp32 pixels / uint8 tiles) and the next ``s_waitcnt`` that names ``vmcnt`` no instruction touches
the load's destination registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _device_asm(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "unused.o")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def _vregs(text):
    """every VGPR an operand list names: v7 -> {7}, v[4:7] -> {4, 5, 6, 7}"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return regs


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump") or shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="needs the ROCm toolchain (llvm-objdump, clang-offload-bundler)")
@pytest.mark.parametrize("addr,width", [(2, 4), (3, 3)], ids=["fp32_pixels", "u8_tiles"])
@pytest.mark.parametrize("unit,mangled_t", [("gemm_bf16", "DF16b"), ("gemm_f16", "DF16_")])
def test_gathered_pixel_registers_are_untouched_until_the_wait(unit, mangled_t, addr, width, tmp_path):
    from plip_amd.build import BUILD, build
    build(verbose=False)
    asm = _device_asm(os.path.join(BUILD, unit + ".o"), str(tmp_path))
    # gemm_nt_kernel<T, 160, 256, 2, 4, EPI_PATCH = 4, SCHED 7, ADDR 2 (fp32 pixels, dwordx4 loads) | 3 (uint8 tiles, dwordx3), NSTAGE 3>
    name = f"_ZN6plipmi14gemm_nt_kernelI{mangled_t}Li160ELi256ELi2ELi4ELi4ELi7ELi{addr}ELi3EEEvNS_10GemmParamsE"
    op = "buffer_load_dwordx%d" % width
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(name), asm, re.S | re.M)
    assert m, f"{name} not found in {unit}.o"
    lines = [l.split("//")[0].strip() for l in m.group(1).splitlines() if l.strip()]
    loads = 0
    for i, ins in enumerate(lines):
        if not ins.startswith(op + " ") or re.search(r"\blds\b", ins):
            continue                                   # LDS-DMA requests have no register destination
        dst = _vregs(ins.split(",")[0])
        assert len(dst) == width, ins
        loads += 1
        for later in lines[i + 1:]:
            if later.startswith("s_waitcnt") and "vmcnt" in later:
                break
            is_load = later.startswith("buffer_load_dwordx")
            if later.startswith("s_") or (is_load and re.search(r"\blds\b", later)):
                continue                               # scalar instructions and LDS-DMA requests name no destination VGPR of ours ...
            touched = _vregs(later.split(None, 1)[1] if " " in later else "")
            if is_load:                                # ... a later pixel load may only share ADDRESS registers, never these
                touched = _vregs(later.split(",")[0])
            assert not (touched & dst), f"{ins!r} is still in flight when {later!r} touches v{sorted(touched & dst)}"
        else:
            pytest.fail(f"no s_waitcnt vmcnt behind {ins!r}")
    # the kernel holds two register sets of five loads (K loop unrolled by two) plus the prologue's: the audit saw them
    assert loads >= 10, loads
