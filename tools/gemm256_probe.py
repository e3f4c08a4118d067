"""k_gemm256.hip (256x256x64, 8 waves, counted vmcnt) against the fp64-accumulating device reference and against the
128x128 kernel of k_gemm16.hip, inside ONE process (q3a_debug_set switches the dispatch):
  1. exactness on ragged / minimal-K / long-K shapes, each repeated (race screen: a stale LDS tile shows as a large error),
  2. TFLOP/s of both kernels on the model's batch-32 encoder / prefill shapes and on 4096^3 / 8192^3.
Usage: python tools/gemm256_probe.py [quick]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import _lib  # noqa: E402
from qwen3_asr_rs_amd.engine import selftest_gemm16  # noqa: E402

lib = _lib.load()


def use256(on: bool, persist: int = 1, group_m: int = 0):
    assert lib.q3a_debug_set(b"gemm256_min_tiles", 0 if on else 1 << 30) == 0
    assert lib.q3a_debug_set(b"gemm256_persist", persist) == 0
    assert lib.q3a_debug_set(b"gemm256_group_m", group_m) == 0


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    bad = 0
    print("== exactness (gemm256 forced) ==")
    use256(True)
    # (the last five: more tiles than CUs -- the persistent walk crosses tile seams; odd / even K-tile counts, K = 128 = two K tiles)
    shapes = [(256, 256, 128), (300, 260, 192), (1000, 480, 128), (513, 770, 896), (12480, 896, 896), (2049, 1024, 3584),
              (390, 2688, 896), (405, 4096, 1024), (12960, 1024, 3072), (1024, 896, 7680),
              (12480, 2688, 896), (12960, 6144, 1024), (8200, 4100, 128), (9000, 3000, 192), (5000, 5000, 448)]
    for shp in shapes:
        for persist in (1, 0):
            use256(True, persist)
            errs = []
            for _ in range(3 if quick else 6):
                r = selftest_gemm16(*shp)
                errs.append(r["err"] / max(r["ref_max"], 1.0))
            ok = max(errs) <= 2e-5
            bad += 0 if ok else 1
            print(f"  {shp} persist={persist}: max rel err over {len(errs)} runs {max(errs):.3e} (min {min(errs):.3e}) {'ok' if ok else 'FAIL'}")
    print("== timing: TFLOP/s old(128x128, 4 waves) vs 256x256 one workgroup per tile (p0) vs 256x256 persistent walk (p1) ==")
    tshapes = [("enc qkv", 12480, 2688, 896), ("enc out", 12480, 896, 896), ("enc fc1", 12480, 3584, 896),
               ("enc fc2", 12480, 896, 3584), ("conv_out", 12480, 896, 7680), ("dec qkv", 12960, 4096, 1024),
               ("dec o", 12960, 1024, 2048), ("dec gate/up", 12960, 6144, 1024), ("dec down", 12960, 1024, 3072),
               ("1.7B qkv B16", 6480, 4096, 2048), ("1.7B down B16", 6480, 2048, 6144),
               ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]
    for name, M, N, K in tshapes:
        res = {}
        for tag, on, persist, gm in (("old", False, 1, 0), ("p0", True, 0, 0), ("p1", True, 1, 0), ("g1", True, 1, 1), ("p0b", True, 0, 0), ("p1b", True, 1, 0), ("g1b", True, 1, 1)):
            use256(on, persist, gm)
            r = selftest_gemm16(M, N, K, reps=10 if K * M * N > 3e11 else 30)
            res[tag] = r["tflops_bf16"]
            if r["err"] > 2e-5 * max(r["ref_max"], 1.0):
                bad += 1
                print(f"  {name}: {tag} WRONG err {r['err']:.3e}")
        us = lambda tf: 2.0 * M * N * K / (tf * 1e12) * 1e6
        print(f"  {name:14s} M={M:6d} N={N:5d} K={K:5d}: old {res['old']:7.1f}   one-per-tile {res['p0']:7.1f} / {res['p0b']:7.1f}   persistent {res['p1']:7.1f} / {res['p1b']:7.1f}   persistent, N-fastest order {res['g1']:7.1f} / {res['g1b']:7.1f} TFLOP/s"
              f"   ({us(max(res['p0'], res['p0b'])):7.1f} -> {us(max(res['p1'], res['p1b'])):7.1f} us)")
    use256(True, 1)
    print("FAILURES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
