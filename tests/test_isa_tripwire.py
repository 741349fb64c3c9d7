"""CPU: static checks on the gfx950 code object inside the built product library (no GPU needed: llvm-objdump only).

Tripwire for the round-2 toolchain finding (tests/gpu_repro/ashr_pk.hip): ROCm 7.2's hipcc fuses "shift, clamp to
0..255, pack" into v_ashr_pk_u8_i32, whose upper destination half is not what the compiler assumes on gfx950; the kernels keep an
opaque barrier behind the shift (enc_kernels.h shr_opaque) where the fusion appeared.  Nothing used to fail if a later edit let the
instruction come back anywhere in the product; now this does."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "h264-lab_amd", "lib", "libh264e_mi355x.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "h264-lab_amd", "csrc"), "all"], stdout=subprocess.DEVNULL)
    d = tmp_path_factory.mktemp("isa")
    shutil.copy(LIB, d / "lib.so")
    subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)     # extracts the code objects next to the file
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(co) == 1, "expected exactly one gfx950 code object in the product library, found %r" % os.listdir(d)
    return subprocess.run([OBJDUMP, "-d", co[0]], cwd=d, capture_output=True, text=True, check=True).stdout


def test_no_packed_shift_clamp_instruction(disassembly):
    bad = [l for l in disassembly.splitlines() if "v_ashr_pk_u8_i32" in l]
    assert not bad, "hipcc emitted v_ashr_pk_u8_i32 (%d sites), e.g. %s -- wrong results on gfx950, see DESIGN.md 4.1; put shr_opaque() behind the shift" % (len(bad), bad[0].strip())


def test_product_is_wave64_gfx950_with_the_expected_kernels(disassembly):
    assert "h264e_mb_kernel" in disassembly and "h264e_synth_kernel" in disassembly and "h264e_ssd_kernel" in disassembly
    assert "v_sad_u8" in disassembly            # the SAD paths really are v_sad_u8
    assert "_dpp" in disassembly                # the row reductions really are DPP
