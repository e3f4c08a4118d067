"""The device code of the product library must not contain a packed fp32 instruction with an op_sel bit set (low result half
reading a HIGH source register): on gfx950 that form returned wrong low halves in lanes 48-63 while a second HIP queue was busy
(csrc/dev.h, DESIGN.md section 8, profiles/r4_pk_op_sel_hazard.txt).  qwen3_asr_rs_amd/build.py keeps hipcc's device assembly
next to the objects and refuses to link when the scan finds one; this test runs the same scan (CPU only: hipcc cross-compiles)."""
import os

import pytest

from qwen3_asr_rs_amd import build

# the two tests that read the product's device assembly cross-compile it first: they need hipcc (present in the build
# container and on the GPU boxes; a box without it skips them instead of failing on a missing .s file)
needs_hipcc = pytest.mark.skipif(not build.have_hipcc(), reason=f"{build.HIPCC} not found: the device assembly cannot be produced here")


def test_scanner_flags_the_hazard_form_and_nothing_else(tmp_path):
    p = tmp_path / "sample.s"
    p.write_text(
        "_ZN3q3a6kernelEv: ; @_ZN3q3a6kernelEv\n"
        "\tv_pk_mul_f32 v[8:9], v[8:9], v[2:3] op_sel:[0,1] op_sel_hi:[0,0]\n"      # the instruction that failed (swap by op_sel)
        "\tv_pk_fma_f32 v[72:73], v[22:23], v[66:67], v[18:19] op_sel:[0,1,0]\n"      # high-register broadcast (conv1 before the fix)
        "\tv_pk_fma_f32 v[18:19], v[32:33], v[66:67], v[16:17] op_sel_hi:[0,1,1]\n"   # low-register broadcast: fine
        "\tv_pk_mul_f32 v[2:3], v[6:7], v[2:3]\n"                                     # straight: fine
        "\tv_pk_add_f32 v[2:3], v[6:7], v[2:3] op_sel:[0,0] op_sel_hi:[1,1]\n"        # explicit defaults: fine
        "\tv_pk_fma_f16 v2, v6, v2, v3 op_sel:[0,1,0]\n")                             # 16-bit packed ops select halves of ONE register: not this hazard
    got = build.scan_isa([str(p)])
    assert [g[2].split()[0] for g in got] == ["v_pk_mul_f32", "v_pk_fma_f32"]
    assert all(g[1] == "_ZN3q3a6kernelEv" for g in got)


@needs_hipcc
def test_product_device_code_has_no_packed_fp32_op_sel():
    build.build(verbose=False)  # incremental; raises by itself if the scan finds the form
    paths = [build.isa_path(s) for s in build.SOURCES if s.endswith(".hip")]
    assert len(paths) >= 13 and all(os.path.exists(p) and os.path.getsize(p) > 1000 for p in paths), paths
    assert build.scan_isa(paths) == []
    assert "-fno-slp-vectorize" in build.FLAGS
    # the scan really reads kernels: the rope kernel is in there, written with single VALU instructions
    txt = open(build.isa_path("k_decode.hip")).read()
    assert "qknorm_rope_kv_kernel" in txt and "v_pk_mul_f32" not in txt.split("qknorm_rope_kv_kernel")[1].split("s_endpgm")[0]


@needs_hipcc
def test_batched_requests_stay_batched_in_the_device_code():
    """Performance structure that hipcc undid once already (tools/isa_waits.py reads the kept assembly):
      * gemm256's fp32-residual epilogue requests the 32 residual rows of a tile back to back (before: one `s_waitcnt vmcnt(0)`
        behind each of them -- 16 exposed round trips per 64-row pass);
      * the o_proj GEMV's split merge requests m, l and the partial outputs of a chunk in one batch (hipcc sank l / o below the branch
        that only needs m); round 6: for up to 8 splits those requests sit IN FRONT of the weight stream, all in one batch with it;
      * the decode qkv / gate-up GEMV requests its whole weight stream before the first wait.
    No GPU: the assembly is what `python -m qwen3_asr_rs_amd.build` keeps."""
    import importlib.util
    import re
    build.build(verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_waits", os.path.join(root, "tools", "isa_waits.py"))
    iw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(iw)
    ks = {k: iw.compress(iw.tokens(b)) for _, k, b in iw.kernels()}
    assert len(ks) > 100

    def longest_load_run(pattern):
        name = [k for k in ks if re.search(pattern, k)]
        assert len(name) == 1, (pattern, name)
        return max([int(n or 1) for n in re.findall(r"\bG(?:x(\d+))?\b", ks[name[0]])] or [0])

    assert longest_load_run(r"gemm256_kernelILb0ENS0_9DenseA256ELb0EEE") >= 15      # the residual rows of two 32-row passes (16) back to back, twice per tile (round 5: all 32 at once, two 64-row passes)
    assert longest_load_run(r"gemv1_kernelILi1ELi4ELb0ELb1ELi0ELb0EEE") >= 26       # chunked merge (> 8 splits): 4 x (m, l, o0, o1) + ... in one batch
    assert longest_load_run(r"gemv1_kernelILi1ELi4ELb0ELb1ELi4ELb0EEE") >= 14       # <= 4 splits: 2 statistics + 8 partial-row loads, then the 4 weight chunks, no wait between
    assert longest_load_run(r"gemv1_kernelILi1ELi4ELb0ELb1ELi8ELb0EEE") >= 24       # <= 8 splits: 4 + 16 + 4
    assert longest_load_run(r"gemv1_kernelILi2ELi4ELb1ELb0ELi0ELb0EEE") >= 24       # qkv / gate-up at K = 2048: x, norm weight (8 + 8) and the weight rows (8)
    assert longest_load_run(r"gemv1_kernelILi2ELi2ELb1ELb0ELi0ELb1EEE") >= 6        # the same at K = 1024, x and the norm weight once per workgroup through LDS: 1 + 1 + 4
    assert iw.exposed(["G", "W0", "j", "G", "G", "W0", "G", "G", "G", "W0"]) == 2   # the suspects metric itself


@needs_hipcc
def test_no_kernel_of_the_product_uses_scratch_memory():
    """Every kernel of the library keeps its working set in registers and LDS: a kernel that starts to spill VGPRs (scratch =
    private memory behind the vector cache) silently turns a register ring into memory traffic -- the pipelined flash attention
    at three workgroups per CU needed 104 B of scratch and was rejected for it (docs/HISTORY.md, round 5).  Read from the AMDGPU
    kernel metadata at the end of hipcc's device assembly; SGPR spills into VGPR lanes are allowed (no memory behind them)."""
    import re
    build.build(verbose=False)
    seen, offenders, widest = 0, [], 0
    for src in build.SOURCES:
        if not src.endswith(".hip"):
            continue
        txt = open(build.isa_path(src)).read()
        for blk in re.split(r"\n  - (?=\.agpr_count)", txt)[1:]:
            field = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)  # noqa: E731
            if not field("name"):
                continue
            seen += 1
            scratch, vspill = int(field("private_segment_fixed_size").group(1)), int(field("vgpr_spill_count").group(1))
            widest = max(widest, int(field("group_segment_fixed_size").group(1)))
            if scratch or vspill:
                offenders.append((src, field("name").group(1), scratch, vspill))
    assert seen > 250, seen            # the metadata was really parsed (309 kernel instantiations in round 5)
    assert not offenders, offenders
    assert widest <= 160 * 1024        # static LDS of the widest kernel fits a CU
