#!/usr/bin/env python
"""HBM-side traffic of the gemm256 launches of the encoder + prefill at PMC_BATCH x 30 s clips, per GEMM shape, next to the
algorithmic bytes of that launch (A operand once + W once + output, + the fp32 residual rows where the epilogue adds them):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/f -o f -- env PMC_BATCH=32 python tools/pmc_target_enc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/w -o w -- env PMC_BATCH=32 python tools/pmc_target_enc.py
    python tools/gemm_traffic.py --fetch OUT/f/..._results.db --write OUT/w/..._results.db [--batch 32]

FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side (Infinity-Cache hits included); on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads (MI355X_MICROARCH.md, HBM section): the table prints the doubled value.  Launches are matched to shapes by
tools/mfma_table.py's classifier (launch order + grid + kernel name; set Q3A_GEMM256_PERSIST=0 in the environment when the traced run
had it)."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("mfma_table", os.path.join(ROOT, "tools", "mfma_table.py"))
mt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mt)


def per_shape(db, counter, shp):
    per = mt.load_pmc([db])
    disp = sorted(per.values(), key=lambda d: d["start"])
    by, matched, unmatched = mt.classify(disp, shp)
    out = {}
    for lab, ds in by.items():
        v = [d["c"].get(counter) for d in ds if counter in d["c"]]
        if v:
            out[lab] = sum(v) / len(v) * 1024.0
    return out, matched, unmatched


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PMC_BATCH", "32")))
    ap.add_argument("--preset", default="0.6b")
    args = ap.parse_args()
    shp = mt.shapes(args.preset, args.batch)
    f, m1, u1 = per_shape(args.fetch, "FETCH_SIZE", shp)
    w, m2, u2 = per_shape(args.write, "WRITE_SIZE", shp)
    print(f"# FETCH_SIZE pass: {m1} gemm256 launches matched, {u1} unmatched; WRITE_SIZE pass: {m2} matched, {u2} unmatched; walk "
          f"{'off (one workgroup per tile)' if os.environ.get('Q3A_GEMM256_PERSIST', '1') == '0' else 'on'}")
    print(f"{'shape':38s} {'tiles':>6s} {'read alg MB':>12s} {'FETCH x2 MB':>12s} {'ratio':>6s} {'write alg MB':>13s} {'WRITE MB':>9s} {'ratio':>6s}")
    seen = set()
    for s in shp:
        lab = s["label"]
        if lab in seen:
            continue
        seen.add(lab)
        M, N, K = s["Mk"], s["N"], s["K"]
        resid = "(+residual)" in lab
        glu = "gate/up" in lab
        rope = "QK-norm" in lab
        if "conv2" in lab or "conv3" in lab:
            # NHWC input map read through the im2col view: each input element belongs to 9/4 output positions on average (3x3, stride 2);
            # algorithmic = the map once.  OH x OW outputs per image -> (2 OH) x (2 OW - ~) inputs: M * 4 * C elements, about
            a_bytes = M * 4 * (K // 9) * 2
        else:
            a_bytes = M * K * 2
        read_alg = a_bytes + N * K * 2 + (M * N * 4 if resid else 0)
        if resid:
            write_alg = M * N * 4
        elif glu:
            write_alg = M * (N // 2) * 2
        else:
            write_alg = M * N * 2  # bf16 outputs (the fused qkv epilogue writes q + the new K / V rows: the same count)
        fx, wx = f.get(lab), w.get(lab)
        fm = lambda v: f"{v / 1e6:12.1f}" if v is not None else f"{'-':>12s}"
        print(f"{lab:38s} {s['tiles']:6d} {read_alg / 1e6:12.1f} {fm(2 * fx if fx is not None else None)} {(2 * fx / read_alg if fx else 0):6.2f} "
              f"{write_alg / 1e6:13.1f} {(wx / 1e6 if wx is not None else 0):9.1f} {(wx / write_alg if wx else 0):6.2f}")


if __name__ == "__main__":
    main()
