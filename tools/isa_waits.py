"""Load / wait structure of the device code, per kernel, from the assembly the build keeps (build/q3asr/*-gfx950.s):
one line of tokens in program order --
  G global load   D LDS-DMA load   S global store   r / w LDS read / write   M MFMA   B barrier   j branch   W<n> s_waitcnt vmcnt(n)
(runs are compressed: Gx16).  What to look for: `G W0 G W0 ...` = every load is an exposed round trip (hipcc puts a load that sits
under a uniform `if (ptr)` into a block of its own with s_waitcnt vmcnt(0) behind it, and SINKS loads to their first use across
branches); `Gx32 W24 W23 ...` = a batch in flight with counted waits.  Round 4 found the fp32-residual epilogue of gemm256 (16
exposed round trips per pass), the fused rope epilogue and the o_proj split merge this way.

    python tools/isa_waits.py [regex over the mangled kernel name]      (no GPU needed; run python -m qwen3_asr_rs_amd.build first)
    python tools/isa_waits.py --suspects    kernels ranked by the number of single loads that are waited for with vmcnt(0) at once"""
import glob, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels():
    for f in sorted(glob.glob(os.path.join(ROOT, "build", "q3asr", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        s = open(f).read()
        for m in re.finditer(r"^(_Z\w+):\s*(;.*)?\n", s, re.M):
            body = s[m.end():]
            end = body.find(".Lfunc_end")
            if end < 0 or ".amdhsa_kernel " + m.group(1) not in s:
                continue
            yield os.path.basename(f).split("-hip-")[0], m.group(1), body[:end]


def tokens(body):
    seq = []
    for l in body.split("\n"):
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        if op.startswith("s_waitcnt"):
            mm = re.search(r"vmcnt\((\d+)\)", l)
            if mm:
                seq.append("W" + mm.group(1))
        elif op.startswith(("global_load", "buffer_load")):
            seq.append("D" if "lds" in l else "G")
        elif op.startswith("global_store"):
            seq.append("S")
        elif op.startswith("s_barrier"):
            seq.append("B")
        elif op.startswith("v_mfma"):
            seq.append("M")
        elif op.startswith(("s_cbranch", "s_branch")):
            seq.append("j")
        elif op.startswith("ds_read"):
            seq.append("r")
        elif op.startswith("ds_write"):
            seq.append("w")
    return seq


def compress(seq):
    out = []
    for t in seq:
        if out and out[-1][0] == t:
            out[-1][1] += 1
        else:
            out.append([t, 1])
    return " ".join(t if n == 1 else f"{t}x{n}" for t, n in out)


def exposed(seq):
    """Loads waited for alone: a run of 1-2 G directly followed (branches aside) by W0."""
    n, i = 0, 0
    core = [t for t in seq if t != "j"]
    while i < len(core):
        if core[i] == "G":
            j = i
            while j < len(core) and core[j] == "G":
                j += 1
            if j - i <= 2 and j < len(core) and core[j] == "W0":
                n += 1
            i = j
        else:
            i += 1
    return n


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--suspects":
        rows = sorted(((exposed(tokens(b)), f, k) for f, k, b in kernels()), reverse=True)
        for n, f, k in rows:
            if n >= 3:
                print(f"{n:4d}  {f:12s} {k}")
    else:
        pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
        for f, k, b in kernels():
            if pat.search(k):
                print("==", f, k)
                print(compress(tokens(b)))
