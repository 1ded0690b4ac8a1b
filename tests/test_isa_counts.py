"""Build-time check of the hand-counted `s_waitcnt vmcnt(N)` schedules in csrc/pv_mlp.hip.

The row-resident kernels order their LDS-DMA weight stream with vmcnt immediates that COUNT the vector-memory operations
a wave issues per block: the DMA pieces (inline asm, invisible to the compiler's own waitcnt pass) and the stores of the
epilogue.  The counts are written in the source as "two stores per block" (ln_linear_rows_kernel) / "four stores per
block" (linear_res_rows_kernel); a compiler that merged, split or re-ordered those stores would make the waits too lax and
MFMAs would read half-landed weights with no signal but numeric drift.  This test compiles the file to gfx950 assembly
(hipcc cross-compiles without a GPU) and counts what the compiler actually emitted.  ROCm 7.2 / hipcc of this image."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def mlp_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    out = str(tmp_path_factory.mktemp("isa") / "pv_mlp.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "pytorchvideo_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "pytorchvideo_amd", "csrc", "pv_mlp.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def _body(asm, mangled_fragment):
    m = re.search(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)s_endpgm" % re.escape(mangled_fragment), asm, re.S | re.M)
    assert m, "kernel %s not found in the assembly" % mangled_fragment
    return m.group(2)


def _meta(asm, mangled_fragment):
    m = re.search(r"\.name:\s+(\S*%s\S*)\n(.*?)\.wavefront_size" % re.escape(mangled_fragment), asm, re.S)
    assert m
    return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", m.group(2))}


def test_linear_res_rows_issues_exactly_four_stores_and_twentyfour_mfmas_per_block(mlp_asm):
    body = _body(mlp_asm, "linear_res_rows_kernelILi24ELi12EE")
    assert body.count("v_mfma_f32_32x32x16_bf16") == 12 * 24
    assert len(re.findall(r"global_store_dwordx4", body)) == 12 * 4          # the counted waits say: 4 per block
    assert len(re.findall(r"global_store_dword(x2|x3)? ", body)) == 0        # ... and nothing narrower beside them
    # the DMA pieces: 6 per wave and block + the two prologue stages (the bias pieces are 4-byte DMAs)
    assert len(re.findall(r"global_load_lds_dwordx4", body)) == 12 * 6 + 2 * 6
    # no compiler-made full drain inside the block loop: vmcnt(0) only before the loop and at the very end
    assert len(re.findall(r"s_waitcnt vmcnt\(0\)", body)) <= 2
    meta = _meta(mlp_asm, "linear_res_rows_kernelILi24ELi12EE")
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0


def test_ln_linear_rows_issues_two_stores_per_block(mlp_asm):
    body = _body(mlp_asm, "ln_linear_rows_kernelILi24ELi2EE")
    # one output block per loop iteration (runtime trip count): the loop body holds the 24 MFMAs once and two 16-byte stores
    assert body.count("v_mfma_f32_32x32x16_bf16") == 24
    assert len(re.findall(r"global_store_dwordx4", body)) == 2
    meta = _meta(mlp_asm, "ln_linear_rows_kernelILi24ELi2EE")
    assert meta["private_segment_fixed_size"] == 0


def test_fused_mlp_main_variant_has_no_scratch_and_streams_its_weights_by_dma(mlp_asm):
    meta = _meta(mlp_asm, "mlp_rows_kernelILi24ELi12ELb1ELi1ELi3ELi0EE")
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0
    body = _body(mlp_asm, "mlp_rows_kernelILi24ELi12ELb1ELi1ELi3ELi0EE")
    assert "global_load_lds_dwordx4" in body
