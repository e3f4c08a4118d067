#!/usr/bin/env python
"""A/B of runtime knobs (q3a_debug_set) on ONE engine in ONE process, settings interleaved round by round, PCM resident:

    python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base skinny_glu_hp3=0 dattn_batched_min_wgs=256
    python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base skinny_q=0

Every setting is a comma-separated list of key=value (or the word `base`); keys not named in a setting keep their defaults.  A key
in capitals (Q3A_DATTN_WARM=0) is an ENVIRONMENT variable instead of a q3a_debug_set key -- for round-local experiment switches the
engine reads at every batch set-up; unset when a setting does not name it.  Per
setting: median wall ms per batch, stage times of the last run, decode us per step, and whether the generated ids equal the first
setting's (the knobs here choose between kernels with the same arithmetic; a knob that changes a reduction order may legitimately
move a near-tie)."""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="0.6b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--new-tokens", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("settings", nargs="+")
    args = ap.parse_args()
    import torch
    import bench
    from qwen3_asr_rs_amd import _lib, synthetic
    lib = _lib.load()
    defaults = {}
    parsed = []
    for sset in args.settings:
        kv = {}
        if sset != "base":
            for item in sset.split(","):
                k, v = item.split("=")
                kv[k] = v if k.isupper() else int(v)
        parsed.append((sset, kv))
    env_keys = sorted({k for _, kv in parsed for k in kv if k.isupper()})
    # defaults of every key that some setting touches (restored between settings)
    known = {"skinny_glu_hp3": 1, "dattn_batched_min_wgs": 128, "skinny_q": 1, "eos_run_ahead": 1,
             "decode_group_size": 0, "decode_parallel_groups": 1, "fuse_qkrope": 1, "gemm256_min_tiles": 128, "gemm256_persist": 1, "gemm256_group_m": 0}
    for _, kv in parsed:
        for k in kv:
            if k.isupper():
                continue
            if k not in known:
                raise SystemExit(f"unknown knob {k} (add its default to tools/ab_knobs.py)")
            defaults[k] = known[k]
    _, eng = bench.make_engine(args.preset, None, 0, False, args.new_tokens)
    clips = [synthetic.synthetic_clip(i, args.seconds) for i in range(args.batch)]
    eng.upload_pcm(clips)
    N = args.new_tokens
    res = {name: dict(ms=[], ids=None, stage=None) for name, _ in parsed}

    def apply(kv):
        for k, v in defaults.items():
            assert lib.q3a_debug_set(k.encode(), v) == 0
        for k in env_keys:
            os.environ.pop(k, None)
        for k, v in kv.items():
            if k.isupper():
                os.environ[k] = v
            else:
                assert lib.q3a_debug_set(k.encode(), v) == 0

    for name, kv in parsed:  # warm-up of every setting (graph capture, first-touch)
        apply(kv)
        eng.run_resident(None, 0, N); eng.fetch_ids(N)
    for _ in range(args.rounds):
        for name, kv in parsed:
            apply(kv)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run_resident(None, 0, N)
            ids = eng.fetch_ids(N)
            torch.cuda.synchronize(); res[name]["ms"].append((time.perf_counter() - t0) * 1e3)
            res[name]["ids"] = ids
            res[name]["stage"] = eng.timings()
    apply({})
    eng.close()
    first = parsed[0][0]
    for name, _ in parsed:
        r = res[name]
        ms = sorted(r["ms"])[len(r["ms"]) // 2]
        st = r["stage"]
        same = r["ids"] == res[first]["ids"]
        ndiff = sum(1 for a, b in zip(r["ids"], res[first]["ids"]) if a != b)
        print(json.dumps({"setting": name, "ms_per_batch": round(ms, 3), "all_ms": [round(x, 2) for x in r["ms"]],
                          "audio_s_per_s": round(args.batch * args.seconds / ms * 1e3, 1),
                          "encoder_ms": round(st["encoder_ms"], 3), "prefill_ms": round(st["prefill_ms"], 3), "decode_ms": round(st["decode_ms"], 3),
                          "decode_us_per_step": round(st["decode_ms"] * 1e3 / max(int(st["decode_steps"]), 1), 2),
                          "ids_equal_to_first_setting": same, "utterances_differing": ndiff,
                          "ids_crc32": zlib.crc32(repr(r["ids"]).encode())}), flush=True)  # (compare across processes / Q3A_LIB builds)


if __name__ == "__main__":
    main()
