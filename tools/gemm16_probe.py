import sys; sys.path.insert(0,'.')
from qwen3_asr_rs_amd.engine import selftest_gemm16
for shp in [(128,128,64),(130,200,96),(390,2688,896),(405,4096,1024),(1000,480,4320),(12480,3584,896),(12480,896,3584),(12960,4096,1024),(12960,1024,3072),(8192,8192,8192)]:
    try:
        r=selftest_gemm16(*shp, reps=5 if shp[0]>1000 else 2)
        print(shp, {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items()})
    except Exception as e: print(shp, "EXC", e)
