"""Build libq3asr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m qwen3_asr_rs_amd.build [--force]

Objects go to build/q3asr/ (git-ignored); the shared library to qwen3_asr_rs_amd/lib/ so it travels
with the repo snapshot to the GPU box.  A source-hash stamp makes rebuilds incremental.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "q3asr")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libq3asr_hip.so")
BIN_DIR = os.path.join(HERE, "bin")
CLI_PATH = os.path.join(BIN_DIR, "asr")  # the reference's CLI (src/main.rs) on top of the C ABI

SOURCES = ["engine.cpp", "model.cpp", "k_gemm.hip", "k_mel.hip", "k_conv1.hip", "k_norm.hip", "k_attn.hip", "k_decode.hip", "k_gemv.hip", "k_dattn.hip", "k_fattn.hip", "k_skinny.hip", "k_gemm16.hip", "k_gemm256.hip", "k_peaks.hip",
           "host_audio.cpp", "host_text.cpp", "host_abi.cpp", "group.cpp", "ops.cpp", "k_ops.hip"]
HEADERS = ["dev.h", "kernels.h", "model.h", "json.h", "host.h", "ops.h", "unicode_tables.h", os.path.join("..", "..", "include", "q3asr.h"),
           os.path.join("..", "..", "include", "q3asr_ops.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP vectoriser packs scalar fp32 code into v_pk_*_f32 with op_sel operand swaps, and a packed fp32
# instruction whose low half reads a high register is unsafe on gfx950 while a second queue is busy (csrc/dev.h, DESIGN.md
# section 8); with the pass off, packed fp32 only comes from explicit f32x2_t code, and scan_isa() below checks what was emitted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# .hip sources only: keeps the device assembly (<stem>-hip-amdgcn-amd-amdhsa-gfx950.s) next to the object for scan_isa().  Part
# of the stamp (hashed below with FLAGS), and an object whose assembly is gone counts as stale.
HIP_FLAGS = ["-x", "hip", "-save-temps=obj"]

# A/B builds for kernel experiments: Q3A_BUILD_VARIANT=name Q3A_BUILD_DEFINES="-DX=1 ..." builds
# lib/libq3asr_hip_<name>.so next to the product library (load it with Q3A_LIB=<path>, see _lib.py).
_VARIANT = os.environ.get("Q3A_BUILD_VARIANT", "")
if _VARIANT:
    FLAGS = FLAGS + os.environ.get("Q3A_BUILD_DEFINES", "").split()
    OBJ_DIR = OBJ_DIR + "_" + _VARIANT
    LIB_PATH = os.path.join(LIB_DIR, f"libq3asr_hip_{_VARIANT}.so")
    CLI_PATH = os.path.join(BIN_DIR, f"asr_{_VARIANT}")


def _hash(paths) -> str:
    h = hashlib.sha1()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + HIP_FLAGS).encode())
    return h.hexdigest()


def have_hipcc() -> bool:
    return os.path.exists(HIPCC) and os.access(HIPCC, os.X_OK)


def _compile(src: str, hdr_hash: str, force: bool) -> str:
    spath = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src + ".o")
    stamp = obj + ".stamp"
    want = _hash([spath]) + hdr_hash
    is_hip = src.endswith(".hip")
    # (an up-to-date object whose kept assembly was cleaned away is rebuilt: scan_isa() must see every kernel)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want and (not is_hip or os.path.exists(isa_path(src))):
        return obj
    cmd = [HIPCC] + FLAGS + (HIP_FLAGS if is_hip else []) + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


_LABEL = re.compile(r"^([A-Za-z_][\w$]*):")
_PK_F32 = re.compile(r"^\s*(v_pk_(?:mul|fma|add)_f32)\b.*\bop_sel:\[([01,]+)\]")


def isa_path(src: str) -> str:
    """Device assembly hipcc kept for a .hip source of the product build (build/q3asr/<stem>-hip-amdgcn-amd-amdhsa-gfx950.s)."""
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def scan_isa(paths=None):
    """Packed fp32 instructions with an op_sel bit set (low result half reading a HIGH source register), per kernel.

    Round 5's standalone sweep (tools/pk_hazard.hip -> profiles/r5_pk_hazard.txt) pins the failing set: v_pk_{mul,add,fma}_f32 whose
    op_sel takes source 0 LOW and source 1 HIGH ([0,1], [0,1,0], [0,1,1]; any op_sel_hi) return the low half as if source 1 were 0 in
    lanes 48-63 whenever a 16-bit-input MFMA (16x16x32 bf16 / f16, 32x32x16 bf16) is in flight on the same CU -- from another queue
    or from other waves of the same kernel.  op_sel_hi-only forms (the only ones the library contains), the other op_sel patterns,
    v_pk_mov_b32, packed f16, DPP and permlane forms never failed.  The scan flags ANY op_sel bit on the three packed-fp32
    arithmetic instructions gfx950 has: a superset of the failing set.
    Returns [(file, kernel, instruction line)]; build() refuses a library that has any."""
    if paths is None:
        paths = [isa_path(s) for s in SOURCES if s.endswith(".hip")]
    found = []
    for path in paths:
        kernel = "?"
        if not os.path.exists(path):
            raise RuntimeError(f"scan_isa: {path} is missing -- rebuild (`python -m qwen3_asr_rs_amd.build`): the device assembly is kept by "
                               "-save-temps=obj and an object without it is treated as stale")
        with open(path) as f:
            for line in f:
                lm = _LABEL.match(line)
                if lm:
                    kernel = lm.group(1)
                m = _PK_F32.match(line)
                if m and "1" in m.group(2):
                    found.append((os.path.basename(path), kernel, line.strip()))
    return found


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hdr_hash = _hash([os.path.join(CSRC, h) for h in HEADERS])
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_hash, force), SOURCES))
    hazards = scan_isa()
    if hazards and os.environ.get("Q3A_ALLOW_ISA_HAZARD", "0") in ("", "0"):
        lines = "\n".join(f"  {f}: {k}: {i}" for f, k, i in hazards[:20])
        raise RuntimeError(f"{len(hazards)} packed fp32 instruction(s) with an op_sel bit set in the device code (csrc/dev.h: unsafe on gfx950; "
                           f"Q3A_ALLOW_ISA_HAZARD=1 builds experiment variants anyway):\n{lines}")
    link_stamp = LIB_PATH + ".stamp"
    want = _hash(objs)
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(link_stamp) or open(link_stamp).read() != want:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(link_stamp, "w") as f:
            f.write(want)
        if verbose:
            print(f"[q3asr build] linked {LIB_PATH}")
    elif verbose:
        print(f"[q3asr build] up to date: {LIB_PATH}")
    # CLI binary: plain C++ over the C ABI, finds the library through $ORIGIN/../lib
    os.makedirs(BIN_DIR, exist_ok=True)
    cli_src = os.path.join(CSRC, "asr_main.cpp")
    cli_stamp = CLI_PATH + ".stamp"
    cli_want = _hash([cli_src, os.path.join(ROOT, "include", "q3asr.h")]) + want
    if force or not os.path.exists(CLI_PATH) or not os.path.exists(cli_stamp) or open(cli_stamp).read() != cli_want:
        cmd = ["g++", "-O2", "-std=c++17", "-o", CLI_PATH, cli_src, "-L" + LIB_DIR, "-lq3asr_hip", "-Wl,-rpath,$ORIGIN/../lib",
               "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"CLI link failed:\n{r.stdout}\n{r.stderr}")
        with open(cli_stamp, "w") as f:
            f.write(cli_want)
        if verbose:
            print(f"[q3asr build] linked {CLI_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
