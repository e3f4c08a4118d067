"""N engines on ONE GPU sharing one weight arena, one host thread each (the C ABI calls release the GIL): independent request
streams whose launches fill each other's gaps.  Prints audio-seconds/s for 1 .. 4 engines at the given batch size per engine.
Run on the GPU box:  python tools/streams_probe.py [batch] [preset]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import torch
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.engine import HipEngine
from qwen3_asr_rs_amd.distributed import pack_arena_host

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
preset = sys.argv[2] if len(sys.argv) > 2 else "0.6b"
d = synthetic.write_checkpoint(f"/tmp/q3a_ckpt_{preset.replace('.', 'p')}_pipe", preset, seed=0, shards=2 if preset == "1.7b" else 1)
clips = [synthetic.synthetic_clip(i, 30.0) for i in range(B)]
arena = pack_arena_host(d).to("cuda:0")
torch.cuda.synchronize()
NE = 4
engs = [HipEngine(d, 0, max_new_tokens=100, device_arena=(arena.data_ptr(), arena.numel())) for _ in range(NE)]
PER = 4  # batches per engine and measurement


def run(eng, n, out):
    for _ in range(n):
        out.append(eng.transcribe_batch(clips, None, max_new=100, fixed_new_tokens=100))


ref = []
for e in engs:
    run(e, 1, ref)
torch.cuda.synchronize()
for n in range(1, NE + 1):
    outs = [[] for _ in range(n)]
    th = [threading.Thread(target=run, args=(engs[i], PER, outs[i])) for i in range(n)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(x == ref[0] for o in outs for x in o)
    print(f"{n} engine(s) x batch {B}: {dt / (n * PER) * 1e3:8.2f} ms per batch overall, {dt / PER * 1e3:8.2f} ms per batch as a request sees it, "
          f"{30.0 * B * n * PER / dt:8.1f} audio-s/s, ids equal: {same}", flush=True)
for e in engs: e.close()
