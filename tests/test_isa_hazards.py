"""The device code of the product library must not contain a packed fp32 instruction with an op_sel bit set (low result half
reading a HIGH source register): on gfx950 that form returned wrong low halves in lanes 48-63 while a second HIP queue was busy
(csrc/dev.h, DESIGN.md section 8, profiles/r4_pk_op_sel_hazard.txt).  qwen3_asr_rs_amd/build.py keeps hipcc's device assembly
next to the objects and refuses to link when the scan finds one; this test runs the same scan (CPU only: hipcc cross-compiles)."""
import os

from qwen3_asr_rs_amd import build


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


def test_product_device_code_has_no_packed_fp32_op_sel():
    build.build(verbose=False)  # incremental; raises by itself if the scan finds the form
    paths = [build.isa_path(s) for s in build.SOURCES if s.endswith(".hip")]
    assert len(paths) >= 13 and all(os.path.exists(p) and os.path.getsize(p) > 1000 for p in paths), paths
    assert build.scan_isa(paths) == []
    assert "-fno-slp-vectorize" in build.FLAGS
    # the scan really reads kernels: the rope kernel is in there, written with single VALU instructions
    txt = open(build.isa_path("k_decode.hip")).read()
    assert "qknorm_rope_kv_kernel" in txt and "v_pk_mul_f32" not in txt.split("qknorm_rope_kv_kernel")[1].split("s_endpgm")[0]
