#!/usr/bin/env python
"""A/B of the input side of q3a_transcribe_batch_ptrs (SURVEY.md section 8d window: host PCM -> ids on the host).

    python tools/upload_ab.py [--preset 0.6b] [--batch 32] [--steps 4]

The knobs are read once per process (Q3A_UPLOAD_MODE / Q3A_UPLOAD_THREADS / Q3A_UPLOAD_PIECES), so every variant runs in a child
process of its own on the same box, PCM-resident and host-to-host interleaved inside each child.  One line per variant."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np  # noqa: F401
    import torch
    from qwen3_asr_rs_amd import synthetic
    import bench
    _, eng = bench.make_engine(args.preset, None, 0, False, args.new_tokens)
    clips = [synthetic.synthetic_clip(i, args.seconds) for i in range(args.batch)]
    N = args.new_tokens
    eng.upload_pcm(clips)
    eng.run_resident(None, 0, N); eng.fetch_ids(N)
    ref = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
    res, h2h = [], []
    for _ in range(args.steps):
        eng.upload_pcm(clips)  # outside the clock (PCM resident)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run_resident(None, 0, N); ids_r = eng.fetch_ids(N)
        torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ids_h = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
        h2h.append(time.perf_counter() - t0)
        assert ids_h == ref and ids_r == ref
    io = eng.io_timings()
    st = eng.timings()
    eng.close()
    med = lambda v: sorted(v)[len(v) // 2]
    r, h = med(res) * 1e3, med(h2h) * 1e3
    print(json.dumps({"mode": os.environ.get("Q3A_UPLOAD_MODE", "1"), "threads": os.environ.get("Q3A_UPLOAD_THREADS", "auto"),
                      "pieces": os.environ.get("Q3A_UPLOAD_PIECES", "8"), "resident_ms": round(r, 3), "host_to_host_ms": round(h, 3),
                      "delta_ms": round(h - r, 3), "delta_pct": round(100 * (h - r) / r, 2),
                      "audio_s_per_s_h2h": round(args.batch * args.seconds / (h / 1e3), 1),
                      "io": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in io.items()},
                      "mel_ms_incl_upload_wait": round(st["mel_ms"], 3)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="0.6b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--new-tokens", type=int, default=100)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--variants", default="0:-1:8,1:-1:8,1:0:8,1:4:8,1:8:16,1:8:32,1:-1:1")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for v in args.variants.split(","):
        mode, thr, pieces = v.split(":")
        env = dict(os.environ, Q3A_UPLOAD_MODE=mode, Q3A_UPLOAD_PIECES=pieces)
        if int(thr) >= 0:
            env["Q3A_UPLOAD_THREADS"] = thr
        else:
            env.pop("Q3A_UPLOAD_THREADS", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--preset", args.preset, "--batch", str(args.batch),
                            "--seconds", str(args.seconds), "--new-tokens", str(args.new_tokens), "--steps", str(args.steps)],
                           env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else json.dumps({"variant": v, "error": (r.stderr or r.stdout)[-300:]}), flush=True)


if __name__ == "__main__":
    main()
