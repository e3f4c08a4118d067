"""Bisect a run-to-run difference of the batched prefill to one launch: engine A (debug taps + Q3A_DEBUG_LAYER_TAPS=1) repeats
one batch while engine B keeps the GPU busy from a second host thread; when A's last hidden rows differ from the first run,
every per-layer tap (K / V cache, attention output, o_proj residual, norm, SwiGLU, down residual) is compared in launch order.
Run on the GPU box:  Q3A_DEBUG_LAYER_TAPS=1 python tools/bisect_layers.py [runs]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.engine import HipEngine
from qwen3_asr_rs_amd.distributed import pack_arena_host
B, RUNS = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 120
P, MAXCTX = 405, 512  # prompt length of a 30 s clip; cache rows per (sequence, kv head) at max_new_tokens = 16
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_pipe", "0.6b", seed=0)
clips = [synthetic.synthetic_clip(i, 30.0) for i in range(B)]
arena = pack_arena_host(d).to("cuda:0"); torch.cuda.synchronize()
A = HipEngine(d, 0, max_new_tokens=16, debug_taps=True, device_arena=(arena.data_ptr(), arena.numel()))
Bg = HipEngine(d, 0, max_new_tokens=100, device_arena=(arena.data_ptr(), arena.numel()))
ENC_ORDER = ["conv3", "enc_in"] + [f"E{li:02d}_x" for li in range(18)] + ["enc_last", "audio_embeds", "dec_embed"]
ORDER = [f"L{li:02d}_{w}" for li in range(28) for w in (("qkvm",) if os.environ.get("Q3A_DEBUG_SCRATCH_COPY") else ()) + ("qkvs", "k", "v", "attn", "o", "ln2", "act", "x")]
def run(): return A.transcribe_batch(clips, None, max_new=2, fixed_new_tokens=2)
run()
ref_last = A.debug_read("dec_last_hidden").copy()
ref = {k: A.debug_read(k).view(np.uint8).copy() for k in ENC_ORDER + ORDER}
print("reference snapshot:", sum(v.nbytes for v in ref.values()) >> 20, "MiB", flush=True)
stop = False
def load():
    while not stop: Bg.transcribe_batch(clips, None, max_new=100, fixed_new_tokens=100)
th = threading.Thread(target=load); th.start(); time.sleep(0.3)
events = 0
for it in range(RUNS):
    run()
    if (A.debug_read("dec_last_hidden").view(np.uint32) == ref_last.view(np.uint32)).all(): continue
    events += 1
    print(f"run {it}: last hidden rows differ", flush=True)
    enc_hit = False
    for k in ENC_ORDER:  # encoder-side taps first (fp32 / fp32 copies of bf16 maps): which stage, which rows
        cur = A.debug_read(k).view(np.uint8)
        ne = cur != ref[k]
        if ne.any():
            el = np.unique(np.nonzero(ne)[0] // 4)
            cols = {"mel": 1, "conv1": 480, "conv2": 480, "conv3": 480, "audio_embeds": 1024, "dec_embed": 1024}.get(k, 896)
            rows = np.unique(el // cols)
            print(f"   first differing ENCODER-side buffer {k}: {len(el)} elements; rows (of {cols} columns) {rows.min()}..{rows.max()} ({len(rows)} distinct): {rows[:16]}; cols {np.unique(el % cols)[:24]}", flush=True)
            a_, b_ = cur.view(np.float32)[el[:8]], ref[k].view(np.float32)[el[:8]]
            print(f"      now {a_}\n      ref {b_}", flush=True)
            enc_hit = True
            break
    if enc_hit: continue
    for k in ORDER:
        cur = A.debug_read(k).view(np.uint8)
        ne = cur != ref[k]
        if k.endswith(("_k", "_v")):  # cache rows at and beyond the prompt length hold the previous run's decode tokens: not part of the prefill
            ne = ne.reshape(-1, MAXCTX, 256).copy()
            ne[:, P:, :] = False
            ne = ne.reshape(-1)
        if k.endswith(("_qkvs", "_qkvm")):  # only the rows the trailing GEMM writes (672 at 32 x 405 prompt rows)
            ne = ne.reshape(-1, 4096 * 4).copy(); ne[672:, :] = False; ne = ne.reshape(-1)
        if ne.any():
            idx = np.nonzero(ne)[0]
            esz = 4 if k.endswith(("_o", "_x", "_qkvs", "_qkvm")) else 2
            cols = {"qkvm": 4096, "qkvs": 4096, "k": 128, "v": 128, "attn": 2048, "o": 1024, "ln2": 1024, "act": 3072, "x": 1024}[k.split("_")[1]]
            el = np.unique(idx // esz)
            rows, cs = el // cols, el % cols
            print(f"   first differing buffer {k}: {len(el)} elements; rows {rows.min()}..{rows.max()} ({len(np.unique(rows))} distinct), cols {cs.min()}..{cs.max()} ({len(np.unique(cs))} distinct)", flush=True)
            print(f"      rows: {np.unique(rows)[:24]}", flush=True)
            print(f"      cols: {np.unique(cs)[:40]}", flush=True)
            if esz == 2:  # the values of the first wrong row, as floats (bf16 bits << 16)
                r0 = int(rows.min())
                f = lambda buf: (buf.view(np.uint16).reshape(-1, cols)[r0].astype(np.uint32) << 16).view(np.float32)
                a_, b_ = f(cur), f(ref[k])
                print(f"      row {r0} cols 104..127 now : {np.array2string(a_[104:128], precision=4, max_line_width=250)}", flush=True)
                print(f"      row {r0} cols 104..127 ref : {np.array2string(b_[104:128], precision=4, max_line_width=250)}", flush=True)
                print(f"      row {r0} cols  40.. 63 equal: {bool((a_[40:64] == b_[40:64]).all())}; |row| {np.linalg.norm(b_):.3f}", flush=True)
            if not k.endswith(("_qkvs", "_qkvm")): break
    if events >= 3: break
stop = True; th.join()
print(f"{events} event(s) in {it + 1} runs", flush=True)
A.close(); Bg.close()
