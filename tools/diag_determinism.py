import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from qwen3_asr_rs_amd import synthetic, _lib
from qwen3_asr_rs_amd.engine import HipEngine
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
clip = synthetic.synthetic_clip(0, 30.0)
N = 100
lib = _lib.load()
for ring in (1,):  # (rounds 3-5 also ran the two-stage staging here: knob gemm16_ring, gone)
    eng = HipEngine(d, 0, max_new_tokens=N, debug_taps=2)
    outs, lasts = [], []
    for r in range(6):
        ids = eng.transcribe_batch([clip], None, max_new=N, fixed_new_tokens=N)[0]
        outs.append(ids); lasts.append(eng.debug_read("dec_last_hidden").view(np.uint32).copy())
    same_ids = [o == outs[0] for o in outs]
    same_last = [bool((l == lasts[0]).all()) for l in lasts]
    print(f"ring={ring}: ids equal to run 0: {same_ids}; prefill last hidden equal: {same_last}", flush=True)
    for r in range(1, 6):
        if outs[r] != outs[0]:
            k = [i for i in range(N) if outs[r][i] != outs[0][i]]
            print("   run", r, "first diffs at", k[:5])
    # stage API on the same engine
    eng.mel([clip]); eng.encode()
    logits, nxt = eng.prefill([HipEngine.build_prompt(390)])
    T = [int(nxt[0])]
    for s in range(N - 1):
        eng.set_next_tokens([outs[0][s]])
        lg, nx, _ = eng.decode_step()
        T.append(int(nx[0]))
    k = [i for i in range(N) if T[i] != outs[0][i]]
    print(f"ring={ring}: stage API vs run 0: diffs at {k[:8]}", flush=True)
    eng.close()
