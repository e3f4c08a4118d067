#!/usr/bin/env python
"""The MFMA bar, quoted honestly: per GEMM shape of the encoder + prefill at PMC_BATCH x 30 s clips

    counter view     MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): counts the PADDED flops the
                     matrix cores executed (tile padding of M, N, K) per cycle AT WHATEVER CLOCK THE CHIP RAN -- it says how busy
                     the matrix pipes were, not how fast the job went;
    same-pass clock  implied SCLK = (GRBM_GUI_ACTIVE / 8) / kernel duration in the same PMC pass (MI355X nominal 2.4 GHz);
    wall-clock view  algorithmic (un-padded) flops of the rows this launch covers / its duration in an UN-instrumented kernel trace
                     (PMC passes serialise dispatches and run 20-40 % slower), as TFLOP/s and as a fraction of the 2.5 PFLOP/s
                     nominal dense bf16 peak (MI355X_MICROARCH.md).

    python tools/mfma_table.py --pmc OUT/b_results.db [--pmc OUT/a_results.db] --trace OUT/trace_results.db [--batch 32] [--preset 0.6b]

Launches of one kernel that serve several shapes with the same grid (196 tiles: conv_out / out / fc2 / proj1 / proj2; 204: o / down)
are told apart by their position in the engine's fixed launch order (csrc/engine.cpp run_encoder / run_prefill)."""
import argparse
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0
CUS = 256  # MI355X
BM = BN = 256
BK = 64


def short(n):
    n = re.sub(r"q3a::\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0]


def split_rows(M, N, rem_max=96):
    """csrc/k_gemm256.hip split_rows(): leading rows that fill whole rounds of 256 tiles."""
    tm, tn = -(-M // BM), -(-N // BN)
    tiles = tm * tn
    rem = tiles % 256
    if rem_max <= 0 or tiles <= 256 or rem == 0 or rem > rem_max:
        return 0
    rows_m = (tiles - rem) // tn
    M1 = rows_m * BM
    return M1 if rows_m >= 1 and M1 < M else 0


def shapes(preset, B, seconds=30.0):
    """(label, kernel substring, M rows on gemm256, N, K, flops multiplier) in launch order per pass, with repeat counts."""
    from qwen3_asr_rs_amd import synthetic
    cfg = synthetic.PRESETS[preset]
    a, t = cfg["audio_config"], cfg["text_config"]
    D, Fn, C, L = a["d_model"], a["encoder_ffn_dim"], a["downsample_hidden_size"], a["encoder_layers"]
    H, I, DL = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    QD, KVD = t["num_attention_heads"] * t["head_dim"], t["num_key_value_heads"] * t["head_dim"]
    frames = int(round(seconds * 100))
    chunks = -(-frames // 100) * B
    conv = lambda x: (x - 1) // 2 + 1
    H1, W1 = conv(128), conv(100); H2, W2 = conv(H1), conv(W1); H3, W3 = conv(H2), conv(W2)
    T = B * (frames // 100 * 13 + (conv(conv(conv(frames % 100))) if frames % 100 else 0))
    P = T + 15 * B
    seq = []   # (label, kernel_substr, M, N, K, addend)
    seq.append(("conv2 implicit GEMM", "ConvA256", chunks * H2 * W2, C, 9 * C, True))
    seq.append(("conv3 implicit GEMM", "ConvA256", chunks * H3 * W3, C, 9 * C, True))
    seq.append(("conv_out (+pos-emb, gather)", "DenseA256", chunks * W3, D, H3 * C, True))
    for _ in range(L):
        seq += [("enc qkv", "DenseA256", T, 3 * D, D, False), ("enc out (+residual)", "DenseA256", T, D, D, False),
                ("enc fc1 (GELU)", "DenseA256", T, Fn, D, False), ("enc fc2 (+residual)", "DenseA256", T, D, Fn, False)]
    seq += [("proj1 (GELU)", "DenseA256", T, D, D, False), ("proj2", "DenseA256", T, a["output_dim"], D, False)]
    for _ in range(DL):
        seq += [("dec qkv + QK-norm/RoPE/KV epilogue", "DenseA256, true", P, QD + 2 * KVD, H, False), ("dec o (+residual)", "DenseA256", P, H, QD, False),
                ("dec gate/up (SwiGLU)", "gemm256_kernel<true", P, 2 * I, H, False), ("dec down (+residual)", "DenseA256", P, H, I, False)]
    out = []
    for label, ksub, M, N, K, addend in seq:
        M1 = 0 if addend else split_rows(M, N)
        Mk = M1 if M1 else M
        tm, tn = -(-Mk // BM), -(-N // BN)
        Kp = -(-K // BK) * BK
        if (-(-M // BM)) * tn < 128:
            continue  # gemm256_eligible(): fewer than gemm256_min_tiles (128) tiles run on the small-tile kernel instead
        # grid: round 6's persistent walk launches min(tiles, CUs) workgroups (Q3A_GEMM256_PERSIST=0: one per tile)
        grid = tm * tn if os.environ.get("Q3A_GEMM256_PERSIST", "1") == "0" else min(tm * tn, CUS)
        out.append(dict(label=label, ksub=ksub, M=M, Mk=Mk, N=N, K=K, tiles=tm * tn, grid=grid, alg=2.0 * Mk * N * K, padded=2.0 * tm * BM * tn * BN * Kp))
    return out


def load_pmc(paths):
    """[(kernel short name, grid workgroups, dispatch_id, {counter: summed value}, duration_us)] in dispatch order."""
    per = {}
    for path in paths:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(grid_size), max(workgroup_size), min(start), max(end) "
                          "from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
        for k, did, c, v, gs, ws, st, en in rows:
            key = (os.path.basename(path), did)
            e = per.setdefault(key, dict(name=short(k), grid=int(gs) // max(int(ws), 1), start=st, dur=(en - st) / 1e3, c={}))
            e["c"][c] = v
    return per


def load_trace(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z from kernels order by start").fetchall()
    return [dict(name=short(n), grid=(gx * gy * gz) // max(wx * wy * wz, 1), dur=(en - st) / 1e3) for n, st, en, gx, gy, gz, wx, wy, wz in rows]


def classify(dispatches, shp):
    """Assign the gemm256 launches (in start order) to the shape sequence: both are in launch order.  A launch takes the next
    shape (looking at most a layer ahead, cyclically over passes) with the same grid and kernel-name substring; a launch that fits
    none is counted as unmatched and skipped.  Returns ({label: [dispatch, ...]}, launches matched, launches unmatched)."""
    g = [d for d in dispatches if d["name"].startswith("gemm256_kernel")]
    by = defaultdict(list)
    pos, matched, unmatched = 0, 0, 0
    n = len(shp)
    for d in g:
        hit = None
        for look in range(6):
            s = shp[(pos + look) % n]
            if d["grid"] == s["grid"] and s["ksub"] in d["name"]:
                hit = (pos + look) % n
                break
        if hit is None:
            unmatched += 1
            continue
        by[shp[hit]["label"]].append(d)
        pos = (hit + 1) % n
        matched += 1
    return by, matched, unmatched


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--trace", default=None)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PMC_BATCH", "32")))
    ap.add_argument("--preset", default="0.6b")
    args = ap.parse_args()
    shp = shapes(args.preset, args.batch)
    uniq = []
    for s in shp:
        if s["label"] not in [u["label"] for u in uniq]:
            uniq.append(s)
    pmc_by = {}
    for p in args.pmc:
        per = load_pmc([p])
        disp = sorted(per.values(), key=lambda d: d["start"])
        by, passes, left = classify(disp, shp)
        print(f"# {os.path.basename(p)}: {passes} gemm256 launches matched to the encoder + prefill launch sequence, {left} unmatched")
        for lab, ds in by.items():
            e = pmc_by.setdefault(lab, defaultdict(list))
            for d in ds:
                for c, v in d["c"].items():
                    e[c].append(v)
                if "GRBM_GUI_ACTIVE" in d["c"]:
                    e["_dur_with_grbm"].append(d["dur"])
    tr_by = {}
    if args.trace:
        by, passes, left = classify(load_trace(args.trace), shp)
        print(f"# {os.path.basename(args.trace)}: {passes} gemm256 launches matched in the un-instrumented kernel trace, {left} unmatched")
        tr_by = {lab: [d["dur"] for d in ds] for lab, ds in by.items()}
    avg = lambda v: sum(v) / len(v) if v else None
    print(f"# Qwen3-ASR-{args.preset}, {args.batch} x 30 s clips; nominal dense bf16 peak {PEAK_TFLOPS:.0f} TFLOP/s; rows = the part of the GEMM the 256 x 256 kernel runs "
          "(a trailing row block may go to the small-tile kernel, csrc/k_gemm256.hip split_rows)")
    hdr = f"{'shape':38s} {'M x N x K (gemm256 rows)':26s} {'tiles':>6s} {'MfmaUtil':>9s} {'pmc us':>8s} {'SCLK GHz':>9s} {'trace us':>9s} {'alg TF/s':>9s} {'of 2.5PF':>9s} {'padded/alg':>10s} {'GFLOP alg':>10s}"
    print(hdr)
    tot_alg = tot_t = 0.0
    wsum = wflops = 0.0
    for s in uniq:
        lab = s["label"]
        e = pmc_by.get(lab, {})
        busy, grbm = avg(e.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), avg(e.get("GRBM_GUI_ACTIVE", []))
        util = busy / (grbm / 8 * 1024) if busy and grbm else None
        pdur = avg(e.get("_dur_with_grbm", []))
        sclk = (grbm / 8) / (pdur * 1e3) if grbm and pdur else None
        tdur = avg(tr_by.get(lab, []))
        tf = s["alg"] / (tdur * 1e6) if tdur else None
        n_per_pass = sum(1 for x in shp if x["label"] == lab)
        if tdur:
            tot_alg += s["alg"] * n_per_pass
            tot_t += tdur * n_per_pass
        if util is not None:
            wsum += util * s["alg"] * n_per_pass
            wflops += s["alg"] * n_per_pass
        f = lambda v, fmt: (fmt % v) if v is not None else "-"
        print(f"{lab:38s} {('%d x %d x %d' % (s['Mk'], s['N'], s['K'])):26s} {s['tiles']:6d} {f(util, '%.3f'):>9s} {f(pdur, '%.1f'):>8s} {f(sclk, '%.2f'):>9s} "
              f"{f(tdur, '%.1f'):>9s} {f(tf, '%.0f'):>9s} {f(tf / PEAK_TFLOPS if tf else None, '%.3f'):>9s} {s['padded'] / s['alg']:10.3f} {s['alg'] * n_per_pass / 1e9:10.0f}")
    if wflops:
        print(f"# FLOP-weighted MfmaUtil over these launches: {wsum / wflops:.3f}")
    if tot_t:
        print(f"# wall-clock over these launches (un-instrumented trace): {tot_alg / 1e9:.0f} GFLOP algorithmic in {tot_t / 1e3:.2f} ms = {tot_alg / tot_t / 1e6:.0f} TFLOP/s = "
              f"{tot_alg / tot_t / 1e6 / PEAK_TFLOPS:.3f} of the nominal peak")


if __name__ == "__main__":
    main()
