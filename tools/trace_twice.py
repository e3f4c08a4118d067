"""Probe: how long does the first rocprofv3 child of a fresh box need before its kernel durations settle?  (bench.py kernel_trace)"""
import sys, os
sys.path.insert(0, os.getcwd())
import bench
W = int(sys.argv[1]) if len(sys.argv) > 1 else 12
inner = ["--preset", "0.6b", "--batch", "1", "--seconds", "30.0", "--new-tokens", "100", "--steps", "3", "--warmup", str(W)]
for i in range(2):
    t = bench.kernel_trace(inner, warmup=W, steps=3)
    for k in ("gemv1_kernel<2, 2, true, false>", "decode_attn_kernel<2, unsigned short>"):
        print(i, k, "timed avg %.2f" % t[k]["avg_us"], "warm-up passes avg %.2f" % t[k]["warmup_avg_us"], flush=True)
