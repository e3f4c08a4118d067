"""GPU diagnostic: per-stage error of the HIP engine against the fp32 oracle (prints, never asserts).
Usage on the GPU box:  python tools/gpu_diag.py [--preset tiny] [--big]  > gpurun_out/diag.log
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import q3asr_oracle as O  # noqa: E402
from qwen3_asr_rs_amd import synthetic  # noqa: E402
from qwen3_asr_rs_amd.engine import HipEngine, selftest_gemm  # noqa: E402


def err(name, got, ref):
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    if got.shape != ref.shape:
        print(f"  {name:18s} SHAPE MISMATCH got {got.shape} ref {ref.shape}")
        return
    d = np.abs(got - ref)
    rel = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
    print(f"  {name:18s} max_abs={d.max():.3e} mean_abs={d.mean():.3e} rel_l2={rel:.3e} ref_max={np.abs(ref).max():.3e} nan={int(np.isnan(got).sum())}")


def stage_compare(model_dir, clips, precise, steps=4):
    print(f"== stage compare: {model_dir} precise={precise} clips={[len(c) for c in clips]}")
    orc = O.AsrOracle(model_dir)
    res = [orc.transcribe_ids(c, fixed_new_tokens=steps, want_taps=True) for c in clips]
    eng = HipEngine(model_dir, 0, precise=precise, debug_taps=True, max_new_tokens=64)
    mels = eng.mel(clips)
    for b, m in enumerate(mels):
        err(f"mel[{b}]", m, res[b].taps["mel"].numpy())
    embeds = eng.encode()
    cat = lambda key, f: np.concatenate([f(r.taps[key]).numpy().ravel() for r in res])
    err("conv1", eng.debug_read("conv1"), cat("conv1", lambda t: t.permute(0, 2, 3, 1).contiguous()))
    err("conv2", eng.debug_read("conv2"), cat("conv2", lambda t: t.permute(0, 2, 3, 1).contiguous()))
    err("conv3", eng.debug_read("conv3"), cat("conv3", lambda t: t.permute(0, 3, 2, 1).contiguous()))
    for k in ("enc_in", "enc_layer0", "enc_last", "audio_embeds"):
        err(k, eng.debug_read(k), cat(k, lambda t: t))
    for b, e in enumerate(embeds):
        err(f"embeds[{b}]", e, res[b].taps["audio_embeds"].numpy())
    prompts = [HipEngine.build_prompt(r.num_audio_tokens) for r in res]
    for b, p in enumerate(prompts):
        ref_ids, _ = O.build_prompt(res[b].num_audio_tokens)
        assert list(p) == ref_ids, "prompt mismatch"
    logits, nxt = eng.prefill(prompts)
    for k in ("dec_embed", "dec_layer0"):
        err(k, eng.debug_read(k), cat(k, lambda t: t))
    err("dec_last_hidden", eng.debug_read("dec_last_hidden"), cat("dec_last_hidden", lambda t: t))
    for b in range(len(clips)):
        err(f"logits0[{b}]", logits[b], res[b].step_logits[0].numpy())
        top = res[b].step_logits[0].topk(2).values
        print(f"    oracle tok0={res[b].all_step_ids[0]} engine tok0={int(nxt[b])} margin={float(top[0]-top[1]):.4f}")
    # teacher-forced decode: feed the oracle's ids
    for s in range(steps - 1):
        eng.set_next_tokens([r.all_step_ids[s] for r in res])
        lg, nx, dn = eng.decode_step()
        for b in range(len(clips)):
            err(f"logits{s+1}[{b}]", lg[b], res[b].step_logits[s + 1].numpy())
            print(f"    oracle tok={res[b].all_step_ids[s+1]} engine tok={int(nx[b])}")
    # free-running whole path
    ids = eng.transcribe_batch(clips, None, max_new=steps, fixed_new_tokens=steps)
    for b in range(len(clips)):
        print(f"  run_resident ids[{b}] = {ids[b]}  oracle = {res[b].all_step_ids[:steps]}")
    print("  timings:", eng.timings())
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also run the 0.6B-dims timing")
    ap.add_argument("--skip-tiny", action="store_true")
    args = ap.parse_args()
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), "threads", torch.get_num_threads())
    print("== gemm self-tests (max_abs_err, ref_abs_max)")
    for (M, N, K) in [(64, 64, 32), (70, 96, 64), (390, 2688, 896), (405, 4096, 1024), (1000, 480, 4320), (3000, 1024, 3072)]:
        for split in (False, True):
            try:
                e, r = selftest_gemm(M, N, K, split)
                print(f"  gemm M={M} N={N} K={K} split={split}: err={e:.3e} ref_max={r:.3e}")
            except Exception as ex:  # noqa: BLE001
                print(f"  gemm M={M} N={N} K={K} split={split}: EXCEPTION {ex}")
    if not args.skip_tiny:
        d = synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny", "tiny", seed=1)
        clips = [synthetic.synthetic_clip(0, 9.3), synthetic.synthetic_clip(1, 2.17)]
        for precise in (True, False):
            try:
                stage_compare(d, clips, precise)
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                print("EXCEPTION in stage_compare:", ex)
        d2 = synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny_untied", "tiny_untied", seed=2, shards=3)
        try:
            stage_compare(d2, [synthetic.synthetic_clip(2, 4.0)], True)
        except Exception as ex:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print("EXCEPTION:", ex)
    if args.big:
        t0 = time.time()
        d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
        print(f"== 0.6B synthetic checkpoint ready in {time.time()-t0:.1f}s")
        t0 = time.time()
        eng = HipEngine(d, 0, precise=False, max_new_tokens=128)
        print(f"   engine load {time.time()-t0:.1f}s")
        clip = synthetic.synthetic_clip(0, 30.0)
        for use_fixed in (100,):
            eng.upload_pcm([clip])
            for it in range(3):
                t0 = time.time()
                eng.run_resident(None, 0, use_fixed)
                dt = time.time() - t0
                print(f"   run_resident B=1 30s N={use_fixed}: wall {dt*1e3:.1f} ms  timings {eng.timings()}")
            print("   ids head:", eng.fetch_ids(use_fixed)[0][:8])
            print("   decode-step profile:", eng.profile_decode_step())
        for B in (4, 8):
            clips = [synthetic.synthetic_clip(i, 30.0) for i in range(B)]
            eng.upload_pcm(clips)
            for it in range(2):
                t0 = time.time()
                eng.run_resident(None, 0, 100)
                print(f"   run_resident B={B}: wall {(time.time()-t0)*1e3:.1f} ms timings {eng.timings()}")
        eng.close()


if __name__ == "__main__":
    main()
