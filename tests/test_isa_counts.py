"""Build-time check of the hand-counted `s_waitcnt vmcnt(N)` schedules in csrc/pv_mlp.hip.

The row-resident kernels order their LDS-DMA weight stream with vmcnt immediates that COUNT the vector-memory operations
a wave issues per block: the DMA pieces (inline asm, invisible to the compiler's own waitcnt pass) and the stores of the
epilogue.  The counts are written in the source as "two stores per block" (ln_linear_rows_kernel); a compiler that merged,
split or re-ordered those stores would make the waits too lax and MFMAs would read half-landed weights with no signal but
numeric drift.  Round 5 adds the eight-phase GEMM (csrc/pv_gemm9.hip): its main loop must carry the counted waits as written
and no scratch traffic (a spilled register's reload is a `vmcnt(0)` that drains the LDS-DMA pipeline).  This test compiles the file to gfx950 assembly
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


def test_ln_linear_rows_issues_two_stores_per_block(mlp_asm):
    body = _body(mlp_asm, "ln_linear_rows_kernelILi24ELi2EE")
    # one output block per loop iteration (runtime trip count): the loop body holds the 24 MFMAs once and two 16-byte stores
    assert body.count("v_mfma_f32_32x32x16_bf16") == 24
    assert len(re.findall(r"global_store_dwordx4", body)) == 2
    meta = _meta(mlp_asm, "ln_linear_rows_kernelILi24ELi2EE")
    assert meta["private_segment_fixed_size"] == 0


def test_fused_mlp_main_variant_has_no_scratch_and_streams_its_weights_by_dma(mlp_asm):
    meta = _meta(mlp_asm, "mlp_rows16_kernelILi12ELi24ELb1ELi3ELi0EE")
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0
    body = _body(mlp_asm, "mlp_rows16_kernelILi12ELi24ELb1ELi3ELi0EE")
    assert "global_load_lds_dwordx4" in body


@pytest.fixture(scope="module")
def gemm9_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    out = str(tmp_path_factory.mktemp("isa9") / "pv_gemm9.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "pytorchvideo_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "pytorchvideo_amd", "csrc", "pv_gemm9.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def _inner_loop(asm, mangled_fragment):
    """The innermost loop that holds MFMAs (the K-tile-pair loop): the basic blocks LLVM's comments attribute to it."""
    m = re.search(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)^\s*\.size\s" % re.escape(mangled_fragment), asm, re.S | re.M)
    assert m, "kernel %s not found" % mangled_fragment
    lines = m.group(2).split("\n")
    best = None
    for i, l in enumerate(lines):
        if "Inner Loop Header" in l:
            lab = lines[i - 1].split(":")[0].strip()
            key = "Header=" + lab[2:] + " "
            idx = [j for j, x in enumerate(lines) if key in x]
            end = (idx[-1] if idx else i) + 1
            while end < len(lines) and not re.match(r"^\.LBB\d+_\d+:", lines[end]):
                end += 1
            body = lines[i - 1:end]
            if sum("v_mfma" in x for x in body) > 0:
                best = body
    assert best is not None
    return best


@pytest.mark.parametrize("frag,wait", [("gemm_quad_kernelILb1ELb0ELi2EE", 6), ("gemm_quad_kernelILb1ELb0ELi0EE", 8),
                                       ("gemm_quad_kernelILb0ELb0ELi2EE", 6), ("gemm_quad_kernelILb1ELb1ELi2EE", 6)])
def test_eight_phase_gemm_main_loop_carries_the_counted_waits_and_no_scratch(gemm9_asm, frag, wait):
    body = _inner_loop(gemm9_asm, frag)
    # the loop header block up to the tile-crossing branch is the common path: 8 phases = 2 K tiles
    text = "\n".join(body)
    assert len(re.findall(r"s_waitcnt vmcnt\(%d\) lgkmcnt\(0\)" % wait, text)) >= 6      # one counted wait per phase (LLVM rotates the loop by up to two phases)
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", text)) >= 64 - 16                  # (LLVM rotates the loop by a phase or two)
    # scratch (spill reloads: a vmcnt(0) each) only in the rarely taken tile-crossing blocks, never in the loop header's path
    header = []
    for l in body[1:]:
        if re.match(r"^\.LBB", l) or "s_cbranch" in l:
            break
        header.append(l)
    assert not any("scratch_" in l for l in header)
    assert sum("scratch_" in l for l in body) <= 24


@pytest.fixture(scope="module")
def gemm9h_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    out = str(tmp_path_factory.mktemp("isa9h") / "pv_gemm9h.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "pytorchvideo_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "pytorchvideo_amd", "csrc", "pv_gemm9h.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.mark.parametrize("frag", ["gemm_quad_half_kernelILb1ELb0ELb0EE", "gemm_quad_half_kernelILb0ELb0ELb0EE",
                                  "gemm_quad_half_kernelILb1ELb1ELb0EE", "gemm_quad_half_kernelILb1ELb0ELb1EE",
                                  "gemm_quad_half_kernelILb0ELb0ELb1EE"])
def test_half_height_gemm_main_loop_carries_the_counted_waits_and_no_scratch(gemm9h_asm, frag):
    """csrc/pv_gemm9h.hip: three K tiles per loop trip, two phases each: vmcnt(6) in phase 0, vmcnt(5) in phase 1, 16 MFMAs
    and 6 LDS-DMAs per K tile, and no scratch anywhere in the kernel (64 accumulator registers leave room)."""
    body = _inner_loop(gemm9h_asm, frag)
    text = "\n".join(body)
    assert len(re.findall(r"s_waitcnt vmcnt\(6\) lgkmcnt\(0\)", text)) >= 2
    assert len(re.findall(r"s_waitcnt vmcnt\(5\) lgkmcnt\(0\)", text)) >= 2
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", text)) >= 48 - 16
    assert len(re.findall(r"global_load_lds_dwordx4", text)) >= 18 - 6
    m = re.search(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)^\s*\.size\s" % re.escape(frag), gemm9h_asm, re.S | re.M)
    assert "scratch_" not in m.group(2)


@pytest.fixture(scope="module")
def attn64_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    out = str(tmp_path_factory.mktemp("isa_attn") / "pv_attn64.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "pytorchvideo_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "pytorchvideo_amd", "csrc", "pv_attn64.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def _loop_with_mfmas(asm, mangled_fragment):
    """Every basic block LLVM attributes to the kernel's loop (label on the header line itself: `.LBBn_m: ; =>This Inner Loop
    Header`), cold blocks placed behind the back edge included."""
    m = re.search(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)^\s*\.size\s" % re.escape(mangled_fragment), asm, re.S | re.M)
    assert m, "kernel %s not found" % mangled_fragment
    lines = m.group(2).split("\n")
    for i, l in enumerate(lines):
        if "Inner Loop Header" in l and l.startswith(".LBB"):
            lab = l.split(":")[0][2:]            # BBn_m
            idx = [j for j, x in enumerate(lines) if ("Header=" + lab + " ") in x]
            end = (idx[-1] if idx else i) + 1    # (a single-block loop has no "in Loop" lines)
            while end < len(lines) and not re.match(r"^\.LBB\d+_\d+:", lines[end]):
                end += 1
            body = lines[i:end]
            if sum("v_mfma" in x for x in body) > 0:
                return body
    raise AssertionError("no loop with MFMAs in %s" % mangled_fragment)


def _num_agpr(asm, mangled_fragment):
    m = re.search(r"\.set\s+\S*%s\S*\.num_agpr,\s*(\d+)" % re.escape(mangled_fragment), asm)
    assert m
    return int(m.group(1))


def test_attention_one_wave_form_keeps_its_tile_loop_free_of_accumulator_copies_and_scratch(attn64_asm):
    """csrc/pv_attn64.hip, the form with 64 query rows per wave (NQB = 2, one wave per SIMD, ~500 registers): what round 6 had
    to fight for, asserted on the emitted code -- the S^T MFMAs write arithmetic registers (inline assembly), O^T stays in
    the accumulator half, and the two tile steps of the loop carry NO v_accvgpr_read/write (each would be an issue slot of an
    issue-bound loop: the first version spent 128 of them per step) and no scratch access; a step holds 48 MFMAs, its 64
    exponentials and 24 transpose reads of V."""
    frag = "attn_w64_kernelILi96ELi2ELi4ELi0EE"
    meta = _meta(attn64_asm, frag)
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0
    assert _num_agpr(attn64_asm, frag) >= 96 + 48  # O^T (96) and Q (48) at least
    body = _loop_with_mfmas(attn64_asm, frag)
    # the common path: everything of the loop outside the cold blocks (the rescale block and the ragged-tile mask are placed
    # behind the loop's back edge by __builtin_expect; they are the only places that may touch accumulator registers by hand)
    text = "\n".join(body)
    hot = text
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", hot)) >= 96
    assert len(re.findall(r"ds_read_b64_tr_b16", hot)) >= 48
    assert len(re.findall(r"v_exp_f32", hot)) >= 128
    assert "scratch_" not in text
    # accumulator copies only inside the rescale block (4 reads + 4 writes per group of four registers, 24 groups per q block)
    n_copy = len(re.findall(r"v_accvgpr_(read|write)_b32", text))
    assert n_copy <= 2 * 2 * 96 * 2, n_copy        # two steps x two q blocks x 96 registers x (read + write), all in cold blocks
    # ... and none between a step's first MFMA and its barrier (the straight-line phase 1)
    for seg in re.split(r"s_barrier", text)[:-1]:
        tail = seg.rsplit("s_cbranch", 1)[-1]      # from the last branch before the barrier = phase 1 of a step
        if "v_mfma" in tail:
            assert "v_accvgpr" not in tail


@pytest.mark.parametrize("frag", ["attn_w64_kernelILi96ELi1ELi8ELi0EE", "attn_w64_kernelILi96ELi1ELi4ELi0EE"])
def test_attention_two_wave_forms_use_no_accumulator_registers(attn64_asm, frag):
    """The 256-register forms (32 query rows per wave, two waves per SIMD): a single "a" constraint makes hipcc split the
    wave's 256 registers 128 / 128 and spill the arithmetic half -- they must compile without accumulator registers, and
    their tile loop without scratch."""
    meta = _meta(attn64_asm, frag)
    assert _num_agpr(attn64_asm, frag) == 0 and meta["vgpr_count"] <= 256
    body = "\n".join(_loop_with_mfmas(attn64_asm, frag))
    assert "scratch_" not in body and "v_accvgpr" not in body
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)) >= 48
