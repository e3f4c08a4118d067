#!/usr/bin/env python
"""Benchmark of the Qwen3-ASR hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (log-mel -> audio encoder -> prefill -> greedy decode -> ids on the host)
over one batch of synthetic 30 s clips.  Default workload = BASELINE.json configs[1]: Qwen3-ASR-0.6B, bf16 weights,
1 clip per GPU, 100 new tokens (EOS ignored: the weights are synthetic, so natural EOS never fires; 100 tokens / 30 s is
the speech rate the survey fixes).  Rank 0 prints ONE JSON line.

What is inside the clock
  value               PCM already resident in HBM when the clock starts (the task contract), generated ids fetched to the
                      host inside the clock (q3a_run_resident + q3a_fetch_ids).
  host_to_host.value  q3a_transcribe_batch: host PCM -> H2D -> hot path -> ids on the host (SURVEY.md section 8d window).

roofline (dominant kernel = the one with the largest share of kernel time IN THE TRACED RUN OF THIS WORKLOAD):
  avg_launch_us       per-launch average from `rocprofv3 --kernel-trace --stats` over graph-replayed steps of this very
                      workload, collected by this script in a child process (in situ: the kernel runs inside the real
                      decode step, between its real neighbours) -- the same command whose summary is committed under
                      profiles/.  If rocprofv3 is unavailable the field falls back to the back-to-back microbenchmark
                      and `avg_launch_us_source` says so.
  microbench          q3a_profile_weight_stream: the same kernel launched back to back from a hipGraph between one
                      HIP event pair on the engine's stream (no neighbours, no tracer).
  decode_stage        the whole decode stage from the HIP events that bracket it inside the timed steps:
                      algorithmic bytes per token / time per token.
  traffic             HBM bytes per launch of the dominant kernel from `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` child
                      passes of a shortened run of this workload (gfx950: FETCH_SIZE doubled for wide streaming reads,
                      MI355X_MICROARCH.md section HBM); `traffic_source` names where the number came from.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import shutil
import sqlite3
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = a port of the reference's tch-CPU op sequence), rank 0 at N=1 only
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(model_dir: str, clip: np.ndarray, new_tokens: int, repeats: int = 3) -> dict:
    """The fp32 oracle (the reference's tch-CPU op sequence, inefficiencies included) timed on this box's host cores on
    a bounded sample: one 30 s clip, front end + prefill + a few decode steps, extrapolated linearly to `new_tokens`
    steps.  Median of `repeats` measurements per thread setting; the better thread setting is reported."""
    from oracle import q3asr_oracle as O
    torch.set_grad_enabled(False)
    orc = O.AsrOracle(model_dir)
    secs = len(clip) / 16000.0
    all_cores = torch.get_num_threads()
    best = None
    # libtorch's default (all cores) is what the reference binary would use; a GEMV-bound decode step often runs
    # faster on fewer threads, so the better of {all cores, 16 threads} is reported, with its core count.
    for nt in sorted({all_cores, min(16, all_cores)}, reverse=True):
        torch.set_num_threads(nt)
        orc.transcribe_ids(clip, fixed_new_tokens=2, keep_logits=False)  # untimed: first-touch / thread-pool start
        runs = []
        for _ in range(repeats):
            t0 = time.time(); orc.transcribe_ids(clip, fixed_new_tokens=2, keep_logits=False); t2 = time.time() - t0
            t0 = time.time(); orc.transcribe_ids(clip, fixed_new_tokens=8, keep_logits=False); t8 = time.time() - t0
            t_dec = max((t8 - t2) / 6.0, 0.0)
            t_front = max(t2 - 2 * t_dec, 0.0)
            runs.append((t_front + new_tokens * t_dec, t_front, t_dec))
        runs.sort()
        med = runs[len(runs) // 2]
        if best is None or med[0] < best[0][0]:
            best = (med, nt, [round(secs / r[0], 3) for r in runs])
    torch.set_num_threads(all_cores)
    (total, t_front, t_dec), nt, rtfx_runs = best
    return {"value": round(secs / total, 3), "unit": "audio-seconds/sec", "cores": nt, "kind": "port",
            "runs": rtfx_runs,
            "sample": f"1 clip x {secs:.0f}s, median of {repeats} runs: mel+encoder+prefill {t_front:.2f}s, decode "
                      f"{t_dec*1e3:.1f} ms/token measured over 6 tokens and extrapolated to {new_tokens} tokens; "
                      f"best of {{{all_cores},16}} threads"}


# ----------------------------------------------------------------------------------------------------------------
# rocprofv3 child passes (in-situ kernel durations, PMC traffic)
# ----------------------------------------------------------------------------------------------------------------
def _short_kernel_name(n: str) -> str:
    n = re.sub(r"q3a::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


def _rocprof(extra_args, inner_args, timeout_s: int):
    """Run `rocprofv3 <extra_args> -- python bench.py --inner ...` from /tmp; return the *_results.db path (or raise)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    out_dir = tempfile.mkdtemp(prefix="q3a_rocprof_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    cmd = [exe] + extra_args + ["-d", out_dir, "-o", "q3a", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--inner"] + inner_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
    dbs = glob.glob(os.path.join(out_dir, "**", "*_results.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        raise RuntimeError(f"rocprofv3 rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}")
    return dbs[0], out_dir


def kernel_trace(inner_args, timeout_s=420) -> dict:
    """{short kernel name: {calls, total_us, avg_us}} from a rocprofv3 --kernel-trace --stats child run."""
    db_path, out_dir = _rocprof(["--kernel-trace", "--stats"], inner_args, timeout_s)
    try:
        db = sqlite3.connect(db_path)
        rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name").fetchall()
        res = {}
        for name, calls, tot, avg in rows:
            k = _short_kernel_name(name)
            e = res.setdefault(k, {"calls": 0, "total_us": 0.0})
            e["calls"] += calls
            e["total_us"] += tot
        for e in res.values():
            e["avg_us"] = e["total_us"] / e["calls"]
        return res
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def pmc_bytes(counter: str, kernel_prefix: str, inner_args, timeout_s=420) -> float:
    """Average `counter` value (KiB) per launch of kernels whose short name starts with kernel_prefix."""
    db_path, out_dir = _rocprof(["--pmc", counter, "--kernel-trace"], inner_args, timeout_s)
    try:
        db = sqlite3.connect(db_path)
        rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name",
                          (counter,)).fetchall()
        n = s = 0.0
        for name, cnt, avg in rows:
            if _short_kernel_name(name).startswith(kernel_prefix):
                n += cnt
                s += cnt * avg
        if n == 0:
            raise RuntimeError(f"no {counter} samples for {kernel_prefix}")
        return s / n
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------------
def make_engine(preset, ckpt_dir, local_rank, precise, new_tokens, arena=None):
    from qwen3_asr_rs_amd import synthetic
    from qwen3_asr_rs_amd.engine import HipEngine
    model_dir = ckpt_dir or f"/tmp/q3a_ckpt_{preset.replace('.', 'p')}"
    synthetic.write_checkpoint(model_dir, preset, seed=0, shards=2 if preset == "1.7b" else 1)
    kw = {}
    if arena is not None:
        kw["device_arena"] = (arena.data_ptr(), arena.numel())
    return model_dir, HipEngine(model_dir, local_rank, precise=precise, max_new_tokens=max(new_tokens, 16), **kw)


def algorithmic_bytes(dims, B, P, new_tokens):
    """SURVEY.md section 8(d) per-unit figures from the model dimensions (bf16 weights)."""
    H, I, V = dims.hidden_size, dims.intermediate_size, dims.vocab_size
    QD, KVD = dims.num_q_heads * dims.head_dim, dims.num_kv_heads * dims.head_dim
    QKV = QD + 2 * KVD
    per_layer = 2.0 * (QKV * H + H * QD + 2 * I * H + H * I)
    weights_per_token = dims.dec_layers * per_layer + 2.0 * V * H           # 1.192 GB at 0.6B
    kv_per_ctx_token = dims.dec_layers * 2 * KVD * 2.0                      # 114 688 B at 0.6B
    ctx_avg = P + (new_tokens - 1) / 2.0
    return {"qkv_gateup_gemv_per_launch": (2.0 * QKV * H + 4.0 * I * H) / 2.0,
            "weights_per_token": weights_per_token,
            "decode_per_step": weights_per_token + B * kv_per_ctx_token * ctx_avg}


def timed_region(eng, clips, steps, warmup, new_tokens, sync_all):
    """W untimed + K timed steps, PCM resident, ids fetched inside the clock.  Returns elapsed seconds (this rank)."""
    eng.upload_pcm(clips)  # inputs resident in HBM before the timed region
    for _ in range(warmup):
        eng.run_resident(None, 0, new_tokens)
        eng.fetch_ids(new_tokens)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_resident(None, 0, new_tokens)  # synchronises its stream before returning
        ids = eng.fetch_ids(new_tokens)        # generated ids -> host
    sync_all()
    elapsed = time.perf_counter() - t0
    assert len(ids) == len(clips) and all(len(x) == new_tokens for x in ids)
    return elapsed


def host_to_host(eng, clips, steps, new_tokens):
    eng.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    return time.perf_counter() - t0


def inner_main(args):
    """Child of the rocprofv3 passes: the same workload, a few graph-replayed steps, nothing printed."""
    from qwen3_asr_rs_amd import synthetic
    _, eng = make_engine(args.preset, args.ckpt_dir, 0, args.precise, args.new_tokens)
    clips = [synthetic.synthetic_clip(i, args.seconds) for i in range(args.batch)]
    eng.upload_pcm(clips)
    for _ in range(args.warmup + args.steps):
        eng.run_resident(None, 0, args.new_tokens)
    eng.close()


def extra_leg(preset, B, seconds, new_tokens, steps, warmup, precise):
    """One more single-GPU workload of BASELINE.json `configs` timed the same way (PCM resident, ids fetched)."""
    from qwen3_asr_rs_amd import synthetic
    t_ck = time.time()
    _, eng = make_engine(preset, None, 0, precise, new_tokens)
    clips = [synthetic.synthetic_clip(i, seconds) for i in range(B)]
    elapsed = timed_region(eng, clips, steps, warmup, new_tokens, torch.cuda.synchronize)
    stage = eng.timings()
    P = int(stage["total_prompt_tokens"]) // B
    ab = algorithmic_bytes(eng.dims, B, P, new_tokens)
    dec_us = stage["decode_ms"] * 1e3 / max(int(stage["decode_steps"]), 1)
    eng.close()
    return {"workload": f"Qwen3-ASR-{preset} bf16, batch={B} x {seconds:.0f}s clips, 1 GPU, {new_tokens} new tokens (fixed)",
            "value": round(B * seconds * steps / elapsed, 3), "unit": "audio-seconds/sec", "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "stage_ms": {k: round(float(v), 3) for k, v in stage.items() if k.endswith("_ms")},
            "decode_stage": {"us_per_step": round(dec_us, 2), "bytes_per_step": round(ab["decode_per_step"]),
                             "achieved_GBps": round(ab["decode_per_step"] / dec_us / 1e3, 1),
                             "frac_of_hbm_peak": round(ab["decode_per_step"] / dec_us / 1e3 / HBM_PEAK_GBPS, 4)},
            "setup_s": round(time.time() - t_ck - elapsed, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="clips per GPU")
    ap.add_argument("--preset", default="0.6b")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--new-tokens", type=int, default=100)
    ap.add_argument("--precise", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child passes (roofline falls back to the microbenchmark)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two PMC child passes (roofline.traffic = null)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU legs (configs[2], configs[3])")
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--trace-out", default=None, help="write the per-kernel table of the in-situ rocprofv3 kernel trace to this file")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if args.inner:
        return inner_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched through torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from qwen3_asr_rs_amd import synthetic

    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        model_dir = args.ckpt_dir or f"/tmp/q3a_ckpt_{args.preset.replace('.', 'p')}"
        if rank == 0:
            synthetic.write_checkpoint(model_dir, args.preset, seed=0, shards=2 if args.preset == "1.7b" else 1)
        dist.barrier()
        from qwen3_asr_rs_amd.distributed import broadcast_arena
        arena = broadcast_arena(model_dir, dev, src=0)  # one RCCL broadcast of the weight arena over xGMI
        torch.cuda.synchronize()
        model_dir, eng = make_engine(args.preset, model_dir, local_rank, args.precise, args.new_tokens, arena)
    else:
        model_dir, eng = make_engine(args.preset, args.ckpt_dir, local_rank, args.precise, args.new_tokens)

    B = args.batch
    clips = [synthetic.synthetic_clip(rank * B + i, args.seconds) for i in range(B)]

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed = timed_region(eng, clips, args.steps, args.warmup, args.new_tokens, sync_all)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage = eng.timings()
    h2h = host_to_host(eng, clips, args.steps, args.new_tokens) if rank == 0 else None
    gemv_path = B <= 2
    stream_prof = eng.profile_weight_stream(reps=4) if gemv_path else None  # back-to-back microbenchmark
    dims = eng.dims
    P = int(stage["total_prompt_tokens"]) // B
    ab = algorithmic_bytes(dims, B, P, args.new_tokens)
    eng.close()

    if rank == 0:
        audio_seconds = world * B * args.seconds * args.steps
        dec_us = stage["decode_ms"] * 1e3 / max(int(stage["decode_steps"]), 1)
        inner = ["--preset", args.preset, "--batch", str(B), "--seconds", str(args.seconds)] + (["--precise"] if args.precise else []) \
            + (["--ckpt-dir", args.ckpt_dir] if args.ckpt_dir else [])
        # ---- dominant kernel, in situ ----
        trace, trace_err, dom = None, None, None
        if world == 1 and not args.no_rocprof:
            try:
                trace = kernel_trace(inner + ["--new-tokens", str(args.new_tokens), "--steps", "2", "--warmup", "1"])
                dom = max(trace.items(), key=lambda kv: kv[1]["total_us"])
                if args.trace_out:
                    tot = sum(v["total_us"] for v in trace.values())
                    with open(args.trace_out, "w") as f:
                        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --inner {' '.join(inner)} --new-tokens {args.new_tokens} --steps 2 --warmup 1\n")
                        f.write(f"# 3 graph-replayed passes of the hot path; total kernel time {tot:.1f} us\n")
                        f.write(f"{'kernel':96s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'pct':>6s}\n")
                        for k, v in sorted(trace.items(), key=lambda kv: -kv[1]["total_us"]):
                            f.write(f"{k[:96]:96s} {v['calls']:7d} {v['total_us']:12.1f} {v['avg_us']:9.2f} {100 * v['total_us'] / tot:6.2f}\n")
            except Exception as ex:  # noqa: BLE001
                trace_err = str(ex)[:300]
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
        gemv_name = "gemv1_kernel<2, 2, true, false>"
        if dom is not None and dom[0].startswith("gemv1_kernel<2, 2, true"):
            # decode qkv + gate/up GEMV (RMSNorm fused, weight streaming [+SwiGLU]): 2 launches per layer per token
            kshort, kinfo = dom
            roof.update(kernel=kshort + " (decode qkv + gate/up GEMV)", bytes_per_launch=round(ab["qkv_gateup_gemv_per_launch"]),
                        avg_launch_us=round(kinfo["avg_us"], 3), launches_traced=kinfo["calls"],
                        share_of_kernel_time=round(kinfo["total_us"] / sum(v["total_us"] for v in trace.values()), 4),
                        avg_launch_us_source="rocprofv3 --kernel-trace --stats, child run of this workload (3 graph-replayed steps), in situ",
                        launches_per_token=2 * dims.dec_layers)
        elif dom is not None:
            # batched configurations: report the dominant kernel's name/time; its algorithmic bytes are per-step figures
            kshort, kinfo = dom
            per_step_calls = kinfo["calls"] / (3.0 * max(args.new_tokens - 1, 1))
            roof.update(kernel=kshort, avg_launch_us=round(kinfo["avg_us"], 3), launches_traced=kinfo["calls"],
                        share_of_kernel_time=round(kinfo["total_us"] / sum(v["total_us"] for v in trace.values()), 4),
                        avg_launch_us_source="rocprofv3 --kernel-trace --stats, child run of this workload, in situ",
                        bytes_per_launch=None, launches_per_token=round(per_step_calls, 2))
        elif stream_prof is not None:
            roof.update(kernel=gemv_name + " (decode qkv + gate/up GEMV)", bytes_per_launch=round(stream_prof["bytes_per_launch"]),
                        avg_launch_us=round(stream_prof["avg_us"], 3),
                        avg_launch_us_source="MICROBENCHMARK (q3a_profile_weight_stream: back-to-back launches, HIP events); "
                                             "rocprofv3 child pass unavailable: " + (trace_err or "skipped"),
                        launches_per_token=2 * dims.dec_layers)
        if roof.get("bytes_per_launch") and roof.get("avg_launch_us"):
            roof["achieved"] = round(roof["bytes_per_launch"] / roof["avg_launch_us"] / 1e3, 1)
        else:  # no per-launch byte model for this kernel: fall back to the decode stage as a whole
            roof["achieved"] = round(ab["decode_per_step"] / dec_us / 1e3, 1)
            roof["achieved_scope"] = "decode stage (all kernels of a step), HIP events in the timed steps"
        roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBPS, 4)
        if stream_prof is not None:
            roof["microbench"] = {"avg_launch_us": round(stream_prof["avg_us"], 3),
                                  "achieved": round(stream_prof["bytes_per_launch"] / stream_prof["avg_us"] / 1e3, 1),
                                  "what": "same kernel, back-to-back from a hipGraph, one HIP event pair on the engine's stream"}
        roof["decode_stage"] = {"us_per_step": round(dec_us, 2), "bytes_per_step": round(ab["decode_per_step"]),
                                "achieved": round(ab["decode_per_step"] / dec_us / 1e3, 1),
                                "frac": round(ab["decode_per_step"] / dec_us / 1e3 / HBM_PEAK_GBPS, 4),
                                "what": "weights + KV bytes per decode step / (decode_ms / decode steps), HIP events inside the timed steps"}
        # ---- HBM traffic of the dominant kernel (PMC child passes on a shortened run) ----
        roof["traffic"], roof["traffic_source"] = None, "not collected"
        if dom is not None and roof.get("bytes_per_launch") and not args.no_pmc:
            try:
                short = inner + ["--new-tokens", "6", "--steps", "1", "--warmup", "0"]
                prefix = dom[0].split("<")[0] + "<" + dom[0].split("<")[1][:10] if "<" in dom[0] else dom[0]
                fetch_kib = pmc_bytes("FETCH_SIZE", prefix, short)
                write_kib = pmc_bytes("WRITE_SIZE", prefix, short)
                roof["traffic"] = round((2.0 * fetch_kib + write_kib) * 1024)
                roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes, 1 step x 6 tokens of this "
                                          "workload); FETCH_SIZE x2 for wide streaming reads on gfx950 (MI355X_MICROARCH.md)")
            except Exception as ex:  # noqa: BLE001
                roof["traffic_source"] = "unavailable: " + str(ex)[:200]
        if trace is not None:
            top = sorted(trace.items(), key=lambda kv: -kv[1]["total_us"])[:6]
            roof["top_kernels"] = [{"kernel": k[:90], "calls": v["calls"], "avg_us": round(v["avg_us"], 2)} for k, v in top]

        out = {
            "metric": "audio-seconds/sec (RTFx) Qwen3-ASR-0.6B greedy, 30s clips" if args.preset == "0.6b"
                      else f"audio-seconds/sec (RTFx) Qwen3-ASR-{args.preset} greedy, 30s clips",
            "value": round(audio_seconds / elapsed, 3),
            "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.precise else "bf16x2-split",
            "data": "synthetic (seeded speech-band noise clips; random-init weights in the reference's safetensors layout)",
            "config": {"workload": f"Qwen3-ASR-{args.preset} bf16, batch={B} x {args.seconds:.0f}s synthetic 16kHz clip per GPU, "
                                   f"{args.new_tokens} new tokens (fixed, EOS ignored)",
                       "clips_per_gpu": B, "clip_seconds": args.seconds, "new_tokens": args.new_tokens,
                       "timed_window": "PCM resident in HBM -> generated ids on the host",
                       "parallelism": f"dp{world} (independent utterances per GPU, no data-path collective)"},
            "stage_ms": {k: round(float(v), 3) for k, v in stage.items() if k.endswith("_ms")},
            "host_to_host": {"value": round(B * args.seconds * args.steps / h2h, 3), "ms_per_step": round(h2h / args.steps * 1e3, 3),
                             "what": "q3a_transcribe_batch on rank 0: host PCM -> H2D -> hot path -> ids on the host"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model_dir, clips[0], args.new_tokens)
            except Exception as ex:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": str(ex)}
        if world == 1 and not args.no_extra and args.preset == "0.6b" and B == 1:
            extra = []
            for preset, b in (("0.6b", 32), ("1.7b", 16)):
                try:
                    extra.append(extra_leg(preset, b, args.seconds, args.new_tokens, steps=3, warmup=1, precise=args.precise))
                except Exception as ex:  # noqa: BLE001
                    extra.append({"workload": f"Qwen3-ASR-{preset} batch={b}", "value": None, "error": str(ex)[:300]})
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
