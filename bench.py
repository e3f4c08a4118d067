#!/usr/bin/env python
"""Benchmark of the Qwen3-ASR hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (log-mel -> audio encoder -> prefill -> greedy decode -> ids on the host)
over one batch of synthetic 30 s clips.  Default workload = BASELINE.json configs[1]: Qwen3-ASR-0.6B, bf16 weights,
1 clip per GPU, 100 new tokens (EOS ignored: the weights are synthetic, so natural EOS never fires; 100 tokens / 30 s is
the speech rate the survey fixes).  Rank 0 prints ONE JSON line.

Launching.  N = 1: `python bench.py`.  N > 1: one process per GPU under torch.distributed.run (the driver's command above);
a bare `python bench.py --gpus N` re-executes itself under torch.distributed.run on 127.0.0.1 with a free port, so the
same command shape works at every N.  With N > 1 the line also carries an `extra` leg: BASELINE configs[4]'s per-GPU
workload (Qwen3-ASR-1.7B, 32 clips per GPU, weights shipped with ONE RCCL broadcast of the arena); the q3a_group_* leg (on by default; --no-native-group-leg) adds
the native q3a_group_* path (one process, one host thread per GPU, RCCL called from C++), run as a child process with a time limit so
that neither a crash nor a hang inside it can cost the line.

What is inside the clock
  value               PCM already resident in HBM when the clock starts (the task contract: the PCIe-inclusive rate is never
                      `value`), generated ids fetched to the host inside the clock (q3a_run_resident + q3a_fetch_ids).
  value_host_to_host  = host_to_host.value: q3a_transcribe_batch_ptrs, host PCM -> H2D -> hot path -> ids on the host (SURVEY.md
                      section 8d's window), same engine, same steps; host_to_host.varying_lengths: the same with utterance
                      lengths that change every step (no reuse of the geometry tables).
  precise_mode        the same workload with opts.precise = 1 (the mode whose greedy ids equal the fp32 oracle's exactly).
  measured_peaks      q3a_measure_peaks on this GPU in this process: HBM read / copy / triad GB/s and the library's own
                      256x256x64 bf16 GEMM on 8192^3 ("measure on the box"); every roofline object also
                      carries frac_of_measured_hbm_read next to frac (of the 8 TB/s nominal peak).
  extra[i].roofline   the same object for BASELINE configs[2] and configs[3] (dominant kernel of THAT workload in situ, its
                      algorithmic bytes per launch, PMC traffic).
  multi_gpu (N > 1)   rccl_ranks_seen (a device all-reduce of ones), per-rank value / ms_per_step, the arena broadcast's pack +
                      wire time, wire-only time of a second device-to-device broadcast and its bit-identity.

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
  natural_eos         the reference's own contract (src/inference.rs:160-167): fixed_new_tokens = 0 on a checkpoint whose
                      <|endoftext|> row is planted so that THIS clip stops after exactly `new_tokens` tokens; the stop
                      condition is evaluated on the device and read from pinned memory (no sync inside the loop).
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
def cpu_baseline(model_dir: str, clip: np.ndarray, new_tokens: int, repeats: int = 5) -> dict:
    """The fp32 oracle (the reference's tch-CPU op sequence, inefficiencies included) timed on this box's host cores on
    a bounded sample: one 30 s clip, front end + prefill + 6 decode steps, extrapolated linearly to `new_tokens` steps
    (t(k tokens) = t_front + k * t_step: two runs per measurement).  The thread setting is chosen by one probe run each
    of {all cores, 16 threads} (libtorch's default is all cores; a GEMV-bound decode step is often faster on fewer), then
    the median of `repeats` measurements with the better setting is reported."""
    from oracle import q3asr_oracle as O
    torch.set_grad_enabled(False)
    orc = O.AsrOracle(model_dir)
    secs = len(clip) / 16000.0
    all_cores = torch.get_num_threads()

    def measure():
        t0 = time.time(); orc.transcribe_ids(clip, fixed_new_tokens=2, keep_logits=False); t2 = time.time() - t0
        t0 = time.time(); orc.transcribe_ids(clip, fixed_new_tokens=8, keep_logits=False); t8 = time.time() - t0
        t_dec = max((t8 - t2) / 6.0, 0.0)
        t_front = max(t2 - 2 * t_dec, 0.0)
        return (t_front + new_tokens * t_dec, t_front, t_dec)

    probe = {}
    for nt in sorted({all_cores, min(16, all_cores)}, reverse=True):
        torch.set_num_threads(nt)
        orc.transcribe_ids(clip, fixed_new_tokens=2, keep_logits=False)  # untimed: first-touch / thread-pool start
        probe[nt] = measure()
    nt = min(probe, key=lambda k: probe[k][0])
    torch.set_num_threads(nt)
    runs = sorted([probe[nt]] + [measure() for _ in range(repeats - 1)])
    torch.set_num_threads(all_cores)
    total, t_front, t_dec = runs[len(runs) // 2]
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        phys = None
    return {"value": round(secs / total, 3), "unit": "audio-seconds/sec", "cores": nt, "cores_physical": phys, "cores_logical": os.cpu_count(),
            "kind": "port",
            "runs": [round(secs / r[0], 3) for r in runs],
            "sample": f"1 clip x {secs:.0f}s, median of {len(runs)} measurements: mel+encoder+prefill {t_front:.2f}s, decode "
                      f"{t_dec*1e3:.1f} ms/token measured over 6 tokens and extrapolated to {new_tokens} tokens; "
                      f"thread setting = better of {{{all_cores},16}} in a probe run each"}


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


def kernel_trace(inner_args, timeout_s=420, warmup=0, steps=1) -> dict:
    """{short kernel name: {calls, total_us, avg_us}} from a rocprofv3 --kernel-trace child run of `warmup` + `steps` identical
    passes.  Like the timed region of the bench itself, the statistics leave the warm-up passes out: of every kernel's
    dispatches (ordered by start time) the first warmup / (warmup + steps) are dropped -- the first pass of a fresh process runs
    its decode kernels 5-15 % slower (cold TLB / instruction caches, clocks still ramping: 5.04 vs 4.79 us for the same GEMV on
    the same box in two consecutive child runs, profiles/r4_trace_warmup_note.txt)."""
    db_path, out_dir = _rocprof(["--kernel-trace", "--stats"], inner_args, timeout_s)
    try:
        db = sqlite3.connect(db_path)
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
        per = {}
        for name, st, en in rows:
            per.setdefault(_short_kernel_name(name), []).append((en - st) / 1e3)
        res = {}
        for k, durs in per.items():
            drop = (len(durs) * warmup) // max(warmup + steps, 1) if len(durs) >= warmup + steps else 0
            kept = durs[drop:]
            res[k] = {"calls": len(kept), "total_us": float(sum(kept)), "avg_us": float(sum(kept)) / len(kept),
                      "warmup_avg_us": (float(sum(durs[:drop])) / drop) if drop else None}
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
    synthetic.write_checkpoint(model_dir, preset, seed=0, shards=2 if preset == "1.7b" else 1, embed_scale=synthetic.PEAKED_EMBED_SCALE)
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
            "decode_per_step": weights_per_token + B * kv_per_ctx_token * ctx_avg,
            # per launch of the kernels that dominate the batched decode step (one launch per layer and step):
            "batched_attention_per_launch": B * (kv_per_ctx_token / dims.dec_layers) * ctx_avg,   # K + V rows of every sequence, one layer
            "gate_up_per_launch": 4.0 * I * H}                                                     # gate + up matrices, bf16


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
    """SURVEY.md section 8d's window: host PCM (one pageable numpy array per clip) -> ids on the host, one
    q3a_transcribe_batch_ptrs call per step.  Returns (elapsed seconds, input-side timings of the last call)."""
    eng.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    t0 = time.perf_counter()
    for _ in range(steps):
        ids = eng.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    elapsed = time.perf_counter() - t0
    assert len(ids) == len(clips) and all(len(x) == new_tokens for x in ids)
    io = eng.io_timings()
    # the engine keeps its geometry tables when the utterance lengths repeat, which every repetition above does.  A real
    # stream of requests does not: the same window once more with lengths that change from call to call (each clip 10 ms shorter per
    # step: same frame / token counts within 1, no table reuse).
    t0 = time.perf_counter()
    for k in range(steps):
        ids = eng.transcribe_batch([c[:len(c) - 160 * (k + 1)] for c in clips], None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    varying = time.perf_counter() - t0
    assert len(ids) == len(clips)
    return elapsed, {"stage_ms": round(io["stage_ms"], 3), "h2d_ms": round(io["h2d_ms"], 3), "pieces": io["pieces"], "host_copy_threads": io["threads"],
                     "varying_lengths_ms_per_step": round(varying / steps * 1e3, 3)}


def h2h_record(B, seconds, steps, h2h, io, resident_ms):
    ms = h2h / steps * 1e3
    vms = io.pop("varying_lengths_ms_per_step", None)
    return {"value": round(B * seconds * steps / h2h, 3), "ms_per_step": round(ms, 3),
            "vs_pcm_resident_pct": round(100.0 * (ms - resident_ms) / resident_ms, 2), "input": io,
            "varying_lengths": None if vms is None else {"value": round(B * (seconds - 0.01 * (steps + 1) / 2.0) * 1e3 / vms, 3), "ms_per_step": vms,
                                                         "what": "the same call with utterance lengths that change every step (geometry tables rebuilt and uploaded each time: no cache hit)"},
            "what": "q3a_transcribe_batch_ptrs on rank 0: host PCM (pageable, one buffer per clip) -> pinned staging by host threads -> "
                    "H2D in pieces on a copy stream, log-mel of a piece under the next piece's copy -> hot path -> ids on the host"}


def inner_main(args):
    """Child of the rocprofv3 passes: the same workload, a few graph-replayed steps, nothing printed."""
    from qwen3_asr_rs_amd import synthetic
    _, eng = make_engine(args.preset, args.ckpt_dir, 0, args.precise, args.new_tokens)
    clips = [synthetic.synthetic_clip(i, args.seconds) for i in range(args.batch)]
    eng.upload_pcm(clips)
    for _ in range(args.warmup + args.steps):
        eng.run_resident(None, 0, args.new_tokens)
    eng.close()


def natural_eos_leg(preset, seconds, new_tokens, steps, warmup, precise, fixed_ms):
    """The reference's stop contract timed beside the fixed-N number: one clip, fixed_new_tokens = 0, on a copy of the
    checkpoint whose <|endoftext|> output row makes this clip emit EOS as token `new_tokens` + 1.  The row is planted
    from the ENGINE's own decoder states (debug tap of the lm_head input over a teacher-free stage-API run; pure numpy,
    no oracle), which the graph-replayed run reproduces bit for bit."""
    from qwen3_asr_rs_amd import synthetic
    from qwen3_asr_rs_amd.engine import HipEngine
    eos_dir = f"/tmp/q3a_ckpt_{preset.replace('.', 'p')}_eosbench"
    synthetic.write_checkpoint(eos_dir, preset, seed=0, shards=2 if preset == "1.7b" else 1, embed_scale=synthetic.PEAKED_EMBED_SCALE)
    key = synthetic.output_embedding_key(eos_dir)
    H = synthetic._locate_tensor(eos_dir, key)[1]["shape"][1]
    synthetic.overwrite_row(eos_dir, key, synthetic.ENDOFTEXT_ID, np.zeros(H, dtype=np.float32))  # un-plant a previous run
    clip = synthetic.synthetic_clip(0, seconds)
    cap = new_tokens + 8
    eng = HipEngine(eos_dir, 0, precise=precise, debug_taps=True, max_new_tokens=cap)
    eng.mel([clip]); eng.encode()
    T = eng.num_audio_tokens(len(clip))
    states = []
    eng.prefill([HipEngine.build_prompt(T)], want_logits=False)
    states.append(eng.debug_read("head_in").copy())
    for _ in range(new_tokens):                 # states behind tokens 1 .. new_tokens (the last one must fire)
        eng.decode_step(want_logits=False)
        states.append(eng.debug_read("head_in").copy())
    eng.close()
    normed = synthetic.final_rms_norm(eos_dir, np.stack(states))
    info = synthetic.plant_eos(eos_dir, normed, [i == new_tokens for i in range(len(states))])
    eng = HipEngine(eos_dir, 0, precise=precise, max_new_tokens=cap)
    eng.upload_pcm([clip])
    for _ in range(warmup):
        eng.run_resident(None, 0, new_tokens)
        eng.run_resident(None, cap, 0)
    torch.cuda.synchronize()
    # paired with the fixed-N run of the SAME engine, alternating (the line's own fixed-N number comes from another engine minutes
    # earlier: box drift of +-1 ms would drown the difference)
    elapsed, paired_fixed = 0.0, 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        eng.run_resident(None, 0, new_tokens)
        eng.fetch_ids(new_tokens)
        t1 = time.perf_counter()
        eng.run_resident(None, cap, 0)
        ids = eng.fetch_ids(cap)
        t2 = time.perf_counter()
        paired_fixed += t1 - t0
        elapsed += t2 - t1
    stage = eng.timings()
    eng.close()
    ok = len(ids[0]) == new_tokens
    return {"workload": f"Qwen3-ASR-{preset} bf16, 1 x {seconds:.0f}s clip, natural EOS after {new_tokens} tokens (max_new {cap})",
            "value": round(seconds * steps / elapsed, 3), "unit": "audio-seconds/sec", "ms_per_step": round(elapsed / steps * 1e3, 3),
            "generated_tokens": len(ids[0]), "stopped_at_planned_token": ok,
            "decode_steps_executed": int(stage["decode_steps"]), "decode_steps_needed": new_tokens,
            "vs_fixed_n_ms": round((elapsed - paired_fixed) / steps * 1e3, 3), "fixed_n_same_engine_ms": round(paired_fixed / steps * 1e3, 3),
            "vs_the_lines_fixed_n_ms": round(elapsed / steps * 1e3 - fixed_ms, 3),
            "eos_row_norm": round(info["row_norm"], 2),
            "what": "fixed_new_tokens = 0: stop condition evaluated on the device, polled from pinned host memory, "
                    "eos_run_ahead = 1 graph replay enqueued ahead (no stream synchronisation inside the loop)"}


def roofline_block(preset, B, seconds, new_tokens, precise, ckpt_dir, dims, ab, dec_us, stream_prof, n_trace, do_trace, do_pmc, trace_out=None,
                   measured_read_gbps=None):
    """The `roofline` object of one workload: the kernel with the largest share of kernel time in `n_trace` rocprofv3 --kernel-trace
    child runs of this very workload (in situ; the run with the median average is reported), its algorithmic bytes per launch
    (SURVEY.md section 8d figures from the model dimensions), HBM traffic from two --pmc child passes, the decode stage as a whole."""
    inner = ["--preset", preset, "--batch", str(B), "--seconds", str(seconds)] + (["--precise"] if precise else []) \
        + (["--ckpt-dir", ckpt_dir] if ckpt_dir else [])
    trace, trace_err, dom, dom_runs = None, None, None, []
    if do_trace:
        try:
            # Profiled child processes on one box differ by 5-8 % in the same kernel's average (profiles/r4_trace_warmup_note.txt:
            # 5.00 / 4.64 / 4.64 us for the dominant GEMV in three consecutive child runs, 4.82 / 4.79 / 5.11 on another box; 12
            # warm-up passes inside one child change nothing), while the un-profiled timed region is steady: the headline's child is
            # run three times, the run with the MEDIAN average of the dominant kernel is reported (and written by --trace-out), and
            # all averages are kept in the line.  (The extra legs run it once: their kernels last 5-15 us and differ less.)
            trace_runs = []
            for _ in range(n_trace):
                trace = kernel_trace(inner + ["--new-tokens", str(new_tokens), "--steps", "3", "--warmup", "1"], warmup=1, steps=3)
                trace_runs.append(trace)
            dom_name = max(trace.items(), key=lambda kv: kv[1]["total_us"])[0]
            dom_runs = [round(t[dom_name]["avg_us"], 3) for t in trace_runs if dom_name in t]
            trace = sorted((t for t in trace_runs if dom_name in t), key=lambda t: t[dom_name]["avg_us"])[(len(dom_runs) - 1) // 2]
            dom = (dom_name, trace[dom_name])
            if trace_out:
                tot = sum(v["total_us"] for v in trace.values())
                with open(trace_out, "w") as f:
                    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --inner {' '.join(inner)} --new-tokens {new_tokens} --steps 3 --warmup 1\n")
                    f.write(f"# the 3 timed passes of the hot path (the warm-up pass's dispatches are left out, as in the bench's own timed region); total kernel time {tot:.1f} us\n")
                    f.write(f"{'kernel':96s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'pct':>6s}\n")
                    for k, v in sorted(trace.items(), key=lambda kv: -kv[1]["total_us"]):
                        f.write(f"{k[:96]:96s} {v['calls']:7d} {v['total_us']:12.1f} {v['avg_us']:9.2f} {100 * v['total_us'] / tot:6.2f}\n")
        except Exception as ex:  # noqa: BLE001
            trace_err = str(ex)[:300]
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    gemv_name = "gemv1_kernel<2, 2, true, false>"
    src = f"rocprofv3 --kernel-trace --stats, child run of this workload (1 warm-up pass left out + 3 timed graph-replayed passes), in situ; " \
          f"median of {n_trace} consecutive child run(s) (profiled processes differ by 5-8 % on one box)"
    if dom is not None and dom[0].startswith("gemv1_kernel<2, 2, true"):
        # decode qkv + gate/up GEMV (RMSNorm fused, weight streaming [+SwiGLU]): 2 launches per layer per token
        kshort, kinfo = dom
        roof.update(kernel=kshort + " (decode qkv + gate/up GEMV)", bytes_per_launch=round(ab["qkv_gateup_gemv_per_launch"]),
                    avg_launch_us=round(kinfo["avg_us"], 3), launches_traced=kinfo["calls"],
                    share_of_kernel_time=round(kinfo["total_us"] / sum(v["total_us"] for v in trace.values()), 4),
                    avg_launch_us_source=src, avg_launch_us_of_the_child_runs=dom_runs, launches_per_token=2 * dims.dec_layers)
    elif dom is not None:
        # batched configurations: the dominant kernel is the batched decode attention (KV stream) or a skinny GEMM (weight stream)
        kshort, kinfo = dom
        per_step_calls = kinfo["calls"] / (3.0 * max(new_tokens - 1, 1))
        bpl, what = None, None
        if kshort.startswith("decode_attn_batched_kernel"):
            bpl, what = ab["batched_attention_per_launch"], "K + V cache rows of all sequences of one layer at the average context (SURVEY 8d: 114 688 B per context token and sequence over 28 layers at 0.6B)"
        elif re.match(r"skinny_kernel<false, [23], ", kshort) and abs(per_step_calls - dims.dec_layers) < 0.5:
            bpl, what = ab["gate_up_per_launch"], "gate + up projection matrices of one layer, bf16 (4 x inter x hidden bytes)"
        roof.update(kernel=kshort, avg_launch_us=round(kinfo["avg_us"], 3), launches_traced=kinfo["calls"],
                    share_of_kernel_time=round(kinfo["total_us"] / sum(v["total_us"] for v in trace.values()), 4),
                    avg_launch_us_source=src, avg_launch_us_of_the_child_runs=dom_runs,
                    bytes_per_launch=round(bpl) if bpl else None, launches_per_token=round(per_step_calls, 2))
        if what:
            roof["bytes_per_launch_what"] = what
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
        if trace_err:
            roof["trace_error"] = trace_err
    roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBPS, 4)
    if measured_read_gbps:
        roof["frac_of_measured_hbm_read"] = round(roof["achieved"] / measured_read_gbps, 4)
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
    if dom is not None and roof.get("bytes_per_launch") and do_pmc:
        try:
            short = inner + ["--new-tokens", "6", "--steps", "1", "--warmup", "0"]
            prefix = dom[0].split("<")[0] + "<" + dom[0].split("<")[1][:10] if "<" in dom[0] else dom[0]
            fetch_kib = pmc_bytes("FETCH_SIZE", prefix, short)
            write_kib = pmc_bytes("WRITE_SIZE", prefix, short)
            roof["traffic"] = round((2.0 * fetch_kib + write_kib) * 1024)
            roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes, 1 step x 6 tokens of this "
                                      "workload: contexts 5 % shorter than the timed run's average); FETCH_SIZE x2 for wide streaming reads on gfx950 (MI355X_MICROARCH.md)")
        except Exception as ex:  # noqa: BLE001
            roof["traffic_source"] = "unavailable: " + str(ex)[:200]
    if trace is not None:
        top = sorted(trace.items(), key=lambda kv: -kv[1]["total_us"])[:6]
        roof["top_kernels"] = [{"kernel": k[:90], "calls": v["calls"], "avg_us": round(v["avg_us"], 2)} for k, v in top]
    return roof


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def multi_gpu_command(n_gpus: int, argv, port: int = 0):
    """The command a bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) re-executes itself as:
    one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.join(ROOT, "bench.py")] + list(argv)


def launch_check(args):
    """`python bench.py --gpus N --launch-check`: the launch path of the N > 1 bench without a GPU (CPU tests): re-launch
    under torch.distributed.run, rendezvous on 127.0.0.1, barrier, max over ranks, one JSON line from rank 0."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = multi_gpu_command(args.gpus, sys.argv[1:])
        os.execv(cmd[0], cmd)
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        top = int(t.item())
    else:
        top = 1
    if rank == 0:
        print(json.dumps({"launch_check": {"world": world, "max_over_ranks": top, "master_addr": os.environ.get("MASTER_ADDR", "")}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def two_streams_leg(model_dir, clips, new_tokens, precise, batches=6):
    """NOT the headline (that is one engine, one batch at a time): two engines on the ONE GPU sharing one weight arena, one
    host thread each (the C ABI releases the GIL), the same batch alternately -- while one engine decodes (latency / HBM
    bound) the other's mel + encoder + prefill (MFMA bound) and decode launches fill its gaps.  Each engine still runs the
    named batch size; per-request latency rises, requests per second rise more.  Host-to-host (PCM upload inside the clock)."""
    import threading
    from qwen3_asr_rs_amd.distributed import pack_arena_host
    from qwen3_asr_rs_amd.engine import HipEngine
    arena = pack_arena_host(model_dir).to("cuda:0")
    torch.cuda.synchronize()
    engs = [HipEngine(model_dir, 0, precise=precise, max_new_tokens=max(new_tokens, 16), device_arena=(arena.data_ptr(), arena.numel()))
            for _ in range(2)]
    seconds = sum(len(c) for c in clips) / 16000.0

    def run(eng, n, out):
        for _ in range(n):
            out.append(eng.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens))

    for e in engs:
        run(e, 1, [])
    torch.cuda.synchronize()
    ref = []
    t0 = time.perf_counter(); run(engs[0], batches, ref); torch.cuda.synchronize(); one = (time.perf_counter() - t0) / batches
    outs = [[], []]
    th = [threading.Thread(target=run, args=(engs[i], batches // 2, outs[i])) for i in range(2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); two = (time.perf_counter() - t0) / (2 * (batches // 2))
    same = all(x == ref[0] for x in outs[0] + outs[1])
    for e in engs: e.close()
    return {"workload": f"two engines on one GPU sharing the weight arena, batch={len(clips)} each, {2 * (batches // 2)} batches alternating "
                        "(host to host)", "value": round(seconds / two, 3), "unit": "audio-seconds/sec", "ms_per_batch": round(two * 1e3, 3),
            "one_engine": {"value": round(seconds / one, 3), "ms_per_batch": round(one * 1e3, 3)}, "ids_equal_to_one_engine": same}


def config4_leg(dist, dev, rank, world, local_rank, seconds, new_tokens, steps, warmup, precise):
    """BASELINE configs[4] scaled to this N: Qwen3-ASR-1.7B, 32 clips per GPU, weights packed on rank 0 and shipped with one
    RCCL broadcast of the arena; all ranks time the same window (barrier + max over ranks)."""
    from qwen3_asr_rs_amd import synthetic
    from qwen3_asr_rs_amd.distributed import broadcast_arena
    B = 32
    model_dir = "/tmp/q3a_ckpt_1p7b"
    if rank == 0:
        synthetic.write_checkpoint(model_dir, "1.7b", seed=0, shards=2, embed_scale=synthetic.PEAKED_EMBED_SCALE)
    dist.barrier()
    t0 = time.perf_counter()
    arena = broadcast_arena(model_dir, dev, src=0)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    _, eng = make_engine("1.7b", model_dir, local_rank, precise, new_tokens, arena)
    clips = [synthetic.synthetic_clip(rank * B + i, seconds) for i in range(B)]

    def sync_all():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    elapsed = timed_region(eng, clips, steps, warmup, new_tokens, sync_all)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    stage = eng.timings()
    eng.close()
    del arena
    return {"workload": f"Qwen3-ASR-1.7b bf16, {world} GPUs x batch={B} x {seconds:.0f}s clips ({world * B} clips), {new_tokens} new tokens (fixed)",
            "value": round(world * B * seconds * steps / float(t.item()), 3), "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(float(t.item()) / steps * 1e3, 3),
            "stage_ms_rank0": {k: round(float(v), 3) for k, v in stage.items() if k.endswith("_ms")},
            "arena_broadcast_s": round(t_bcast, 3)}


def native_group_leg(preset, ckpt_dir, world, B, seconds, new_tokens, steps, precise):
    """The same workload through q3a_group_* (one process, one host thread per GPU, ncclBroadcast called from C++,
    csrc/group.cpp), host PCM -> ids on the host: run on rank 0 while the other ranks wait at a barrier."""
    from qwen3_asr_rs_amd import synthetic
    from qwen3_asr_rs_amd.engine import HipGroup
    model_dir = ckpt_dir or f"/tmp/q3a_ckpt_{preset.replace('.', 'p')}"
    t0 = time.perf_counter()
    grp = HipGroup(model_dir, n_gpus=world, precise=precise, max_new_tokens=max(new_tokens, 16))
    t_create = time.perf_counter() - t0
    clips = [synthetic.synthetic_clip(i, seconds) for i in range(world * B)]
    grp.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    t0 = time.perf_counter()
    for _ in range(steps):
        ids = grp.transcribe_batch(clips, None, max_new=new_tokens, fixed_new_tokens=new_tokens)
    elapsed = time.perf_counter() - t0
    out = {"workload": f"q3a_group_transcribe: Qwen3-ASR-{preset} bf16, {world} GPUs x {B} x {seconds:.0f}s clips, host PCM -> ids",
           "value": round(world * B * seconds * steps / elapsed, 3), "unit": "audio-seconds/sec", "ms_per_step": round(elapsed / steps * 1e3, 3),
           "ranks": grp.size, "used_rccl": grp.used_rccl, "group_create_s": round(t_create, 2),
           "startup_s": {k: round(v, 3) for k, v in grp.startup_seconds.items()},
           "stage_ms_per_rank": [{k: round(float(v), 2) for k, v in grp.engine_timings(r).items() if k.endswith("_ms")} for r in range(grp.size)],
           "ids_ok": len(ids) == world * B and all(len(x) == new_tokens for x in ids)}
    grp.close()
    return out


def native_group_child(args, world, B, timeout_s=600):
    """The q3a_group_* leg in a process of its own (`bench.py --inner-group`), bounded by a timeout: this path has never met a
    multi-GPU box, and neither a crash nor a hang inside it may cost the bench line.  Returns the leg's record or an error record."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--inner-group", "--gpus", str(world), "--batch", str(B), "--preset", args.preset,
           "--seconds", str(args.seconds), "--new-tokens", str(args.new_tokens), "--steps", str(args.steps)] \
        + (["--precise"] if args.precise else []) + (["--ckpt-dir", args.ckpt_dir] if args.ckpt_dir else [])
    rec = {"workload": "q3a_group_transcribe", "value": None}
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)  # on timeout: kills this child only
    except subprocess.TimeoutExpired:
        return dict(rec, error=f"no result within {timeout_s} s (child killed)")
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return dict(rec, error=f"rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}")


def inner_group_main(args):
    """Child of native_group_child: one JSON line on stdout."""
    try:
        out = native_group_leg(args.preset, args.ckpt_dir, args.gpus, args.batch, args.seconds, args.new_tokens, args.steps, args.precise)
    except Exception as ex:  # noqa: BLE001
        out = {"workload": "q3a_group_transcribe", "value": None, "error": str(ex)[:300]}
    print(json.dumps(out), flush=True)


def extra_leg(preset, B, seconds, new_tokens, steps, warmup, precise, do_trace=True, do_pmc=True, measured_read_gbps=None, trace_out=None):
    """One more single-GPU workload of BASELINE.json `configs` timed the same way (PCM resident, ids fetched), with its own
    `roofline` object (dominant kernel in situ from one rocprofv3 child run of THIS workload, PMC traffic) -- VERDICT r5 item 2."""
    from qwen3_asr_rs_amd import synthetic
    t_ck = time.time()
    _, eng = make_engine(preset, None, 0, precise, new_tokens)
    clips = [synthetic.synthetic_clip(i, seconds) for i in range(B)]
    elapsed = timed_region(eng, clips, steps, warmup, new_tokens, torch.cuda.synchronize)
    stage = eng.timings()
    P = int(stage["total_prompt_tokens"]) // B
    ab = algorithmic_bytes(eng.dims, B, P, new_tokens)
    dec_us = stage["decode_ms"] * 1e3 / max(int(stage["decode_steps"]), 1)
    h2h, h2h_io = host_to_host(eng, clips, steps, new_tokens)  # SURVEY.md section 8d's window for this workload too
    dims = eng.dims
    eng.close()
    roof = roofline_block(preset, B, seconds, new_tokens, precise, None, dims, ab, dec_us, None, n_trace=1, do_trace=do_trace, do_pmc=do_pmc,
                          trace_out=trace_out, measured_read_gbps=measured_read_gbps)
    return {"workload": f"Qwen3-ASR-{preset} bf16, batch={B} x {seconds:.0f}s clips, 1 GPU, {new_tokens} new tokens (fixed)",
            "value": round(B * seconds * steps / elapsed, 3), "unit": "audio-seconds/sec", "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "stage_ms": {k: round(float(v), 3) for k, v in stage.items() if k.endswith("_ms")},
            "host_to_host": h2h_record(B, seconds, steps, h2h, h2h_io, elapsed / steps * 1e3),
            "decode_stage": {"us_per_step": round(dec_us, 2), "bytes_per_step": round(ab["decode_per_step"]),
                             "achieved_GBps": round(ab["decode_per_step"] / dec_us / 1e3, 1),
                             "frac_of_hbm_peak": round(ab["decode_per_step"] / dec_us / 1e3 / HBM_PEAK_GBPS, 4)},
            "roofline": roof,
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
    ap.add_argument("--no-peaks", action="store_true", help="skip q3a_measure_peaks (measured HBM / MFMA denominators)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (N = 1: configs[2], configs[3], natural EOS; N > 1: configs[4]'s per-GPU workload)")
    ap.add_argument("--no-native-group-leg", action="store_true", help="N > 1: skip the q3a_group_* leg (on by default; its error, if any, is captured in the JSON)")
    ap.add_argument("--native-group-leg", action="store_true", help="(accepted for compatibility: the leg is on by default since round 4)")
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--trace-out", default=None, help="write the per-kernel table of the in-situ rocprofv3 kernel trace to this file")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--inner-group", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true",
                    help="only exercise the N > 1 launch path (self re-launch, rendezvous, barrier, max over ranks) on the gloo backend: no GPU needed")
    args = ap.parse_args()

    if args.launch_check:
        return launch_check(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if args.inner:
        return inner_main(args)
    if args.inner_group:
        return inner_group_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: same command shape as the N = 1 line -> one process per GPU
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        cmd = multi_gpu_command(args.gpus, sys.argv[1:])
        sys.stderr.write("[bench] re-launching as: " + " ".join(cmd) + "\n")
        sys.stderr.flush()
        os.execv(cmd[0], cmd)
    if world != args.gpus:
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
            synthetic.write_checkpoint(model_dir, args.preset, seed=0, shards=2 if args.preset == "1.7b" else 1, embed_scale=synthetic.PEAKED_EMBED_SCALE)
        dist.barrier()
        from qwen3_asr_rs_amd.distributed import broadcast_arena
        # self-validating N > 1 line: how many ranks RCCL itself saw (a device all-reduce of ones), the time of the
        # ONE arena broadcast (rank 0 packs on the host first: pack and wire time apart), and below every rank's own elapsed time
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        t_b0 = time.perf_counter()
        arena = broadcast_arena(model_dir, dev, src=0)  # one RCCL broadcast of the weight arena over xGMI
        torch.cuda.synchronize()
        t_b1 = time.perf_counter()
        warm = torch.empty_like(arena)
        if rank == 0:
            warm.copy_(arena)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t_b2 = time.perf_counter()
        dist.broadcast(warm, src=0)       # the same bytes once more, device to device only: the wire time of the broadcast
        torch.cuda.synchronize()
        t_b3 = time.perf_counter()
        same = bool(torch.equal(warm, arena))
        del warm
        multi_info = {"rccl_ranks_seen": int(ones.item()), "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "arena_bytes": int(arena.numel()), "arena_pack_plus_broadcast_s": round(t_b1 - t_b0, 3),
                      "arena_broadcast_wire_s": round(t_b3 - t_b2, 4), "arena_broadcast_GBps": round(arena.numel() / max(t_b3 - t_b2, 1e-9) / 1e9, 1),
                      "second_broadcast_bit_identical": same}
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
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_s = [float(x.item()) for x in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        multi_info["per_rank_value"] = [round(B * args.seconds * args.steps / x, 3) for x in per_rank_s]
        multi_info["per_rank_ms_per_step"] = [round(x / args.steps * 1e3, 3) for x in per_rank_s]
    stage = eng.timings()
    h2h, h2h_io = host_to_host(eng, clips, args.steps, args.new_tokens) if rank == 0 else (None, None)
    gemv_path = B <= 2
    stream_prof = eng.profile_weight_stream(reps=4) if gemv_path else None  # back-to-back microbenchmark
    dims = eng.dims
    P = int(stage["total_prompt_tokens"]) // B
    ab = algorithmic_bytes(dims, B, P, args.new_tokens)
    eng.close()
    multi_extra = []
    if dist is not None and not args.no_extra:
        try:  # every rank takes part (collectives inside)
            multi_extra.append(config4_leg(dist, dev, rank, world, local_rank, args.seconds, args.new_tokens, 2, 1, args.precise))
        except Exception as ex:  # noqa: BLE001
            multi_extra.append({"workload": "Qwen3-ASR-1.7b, 32 clips per GPU", "value": None, "error": str(ex)[:300]})
        torch.cuda.synchronize()
        dist.barrier()
        if not args.no_native_group_leg:
            # the native one-process group drives every GPU from rank 0's process; the other ranks wait on a gloo (CPU) barrier so
            # that no collective kernel of theirs spins on the GPUs meanwhile
            cpu_pg = dist.new_group(backend="gloo")
            if rank == 0:
                multi_extra.append(native_group_child(args, world, B))
            dist.barrier(group=cpu_pg)

    if rank == 0:
        audio_seconds = world * B * args.seconds * args.steps
        dec_us = stage["decode_ms"] * 1e3 / max(int(stage["decode_steps"]), 1)
        peaks = None
        if world == 1 and not args.no_peaks:
            try:
                from qwen3_asr_rs_amd.engine import measure_peaks
                peaks = measure_peaks(local_rank, 5)
                peaks["what"] = ("q3a_measure_peaks (csrc/k_peaks.hip), this process, this GPU, best of 5: read-only stream over 2 GiB (16 B per lane, 8 loads in "
                                 "flight, non-temporal: the decode weight streams' access pattern), 1 GiB copy and fp32 triad (bytes counted on every "
                                 "stream), the library's own 256x256x64 bf16 GEMM on 8192^3 -- on constant operands (mfma_bf16_TFLOPs: nothing toggles, the clock "
                                 "stays high: the ceiling) and on random ones (mfma_bf16_TFLOPs_random_data: what real data draws); nominal: 8000 GB/s, 2500 TFLOP/s dense bf16")
            except Exception as ex:  # noqa: BLE001
                peaks = {"error": str(ex)[:200]}
        roof = roofline_block(args.preset, B, args.seconds, args.new_tokens, args.precise, args.ckpt_dir, dims, ab, dec_us, stream_prof,
                              n_trace=3, do_trace=(world == 1 and not args.no_rocprof), do_pmc=not args.no_pmc, trace_out=args.trace_out,
                              measured_read_gbps=(peaks or {}).get("hbm_read_GBps"))

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
            "host_to_host": h2h_record(B, args.seconds, args.steps, h2h, h2h_io, elapsed / args.steps * 1e3),
            "roofline": roof,
        }
        # SURVEY 8d's window (host PCM -> ids on the host) beside the contract's `value` (PCM resident in HBM when the clock starts -- the
        # task contract forbids the PCIe-inclusive rate as `value`): same engine, same steps, rank 0
        out["value_host_to_host"] = out["host_to_host"]["value"]
        if peaks is not None:
            out["measured_peaks"] = peaks
        if dist is not None:
            out["multi_gpu"] = multi_info
        if world == 1 and not args.no_extra and B == 1 and not args.precise:
            try:  # the exact-ids mode has a number too: fp32 activations split hi + lo, fp32 KV cache
                _, peng = make_engine(args.preset, args.ckpt_dir, local_rank, True, args.new_tokens)
                pel = timed_region(peng, clips, 3, 1, args.new_tokens, torch.cuda.synchronize)
                pst = peng.timings()
                peng.close()
                out["precise_mode"] = {"value": round(B * args.seconds * 3 / pel, 3), "unit": "audio-seconds/sec", "ms_per_step": round(pel / 3 * 1e3, 3),
                                       "steps": 3, "warmup": 1, "dtype": "bf16 weights, fp32 activations as bf16 hi + lo pairs, fp32 KV cache",
                                       "stage_ms": {k: round(float(v), 3) for k, v in pst.items() if k.endswith("_ms")},
                                       "what": "opts.precise = 1: the mode whose greedy ids equal the fp32 oracle's exactly (tests/test_gpu_parity.py, "
                                               "test_config0: logits <= 2e-4); the timed default mode is exact over the 2 x error margin"}
            except Exception as ex:  # noqa: BLE001
                out["precise_mode"] = {"value": None, "error": str(ex)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model_dir, clips[0], args.new_tokens)
            except Exception as ex:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": str(ex)}
        if world == 1 and not args.no_extra and B == 1:
            try:
                out["natural_eos"] = natural_eos_leg(args.preset, args.seconds, args.new_tokens, args.steps, args.warmup, args.precise,
                                                     elapsed / args.steps * 1e3)
            except Exception as ex:  # noqa: BLE001
                out["natural_eos"] = {"value": None, "error": str(ex)[:300]}
        if world == 1 and not args.no_extra and B == 1:
            try:
                out["two_streams"] = two_streams_leg(model_dir, clips, args.new_tokens, args.precise)
            except Exception as ex:  # noqa: BLE001
                out["two_streams"] = {"value": None, "error": str(ex)[:300]}
        if world == 1 and not args.no_extra and args.preset == "0.6b" and B == 1:
            extra = []
            for preset, b in (("0.6b", 32), ("1.7b", 16)):
                try:
                    tout = (args.trace_out + f".{preset.replace('.', 'p')}_b{b}") if args.trace_out else None
                    extra.append(extra_leg(preset, b, args.seconds, args.new_tokens, steps=3, warmup=1, precise=args.precise,
                                           do_trace=not args.no_rocprof, do_pmc=not args.no_pmc, measured_read_gbps=(peaks or {}).get("hbm_read_GBps"),
                                           trace_out=tout))
                except Exception as ex:  # noqa: BLE001
                    extra.append({"workload": f"Qwen3-ASR-{preset} batch={b}", "value": None, "error": str(ex)[:300]})
            out["extra"] = extra
        if multi_extra:
            out["extra"] = multi_extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
