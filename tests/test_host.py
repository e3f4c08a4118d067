"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/q3asr.h declares,
the host-only entry points (config parse, safetensors loader, arena packer, prompt builder) agree with the
oracle, and the product path fails loudly without a HIP device (no compute is attempted here)."""
import ctypes as C
import json
import os
import re
import struct

import numpy as np
import pytest
import torch

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import _lib, synthetic
from qwen3_asr_rs_amd.engine import HipEngine, Q3aError, capitalize_first, parse_asr_output

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "q3asr.h")).read()
    declared = sorted(set(re.findall(r"\b(q3a_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(_lib.SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), f"libq3asr_hip.so does not export {name}"


def test_debug_keys_are_documented_and_reject_unknown(lib):
    """Every key q3a_debug_set accepts is described in include/q3asr.h (and vice versa); unknown keys are an error, no GPU needed."""
    src = open(os.path.join(ROOT, "qwen3_asr_rs_amd", "csrc", "engine.cpp")).read()
    body = src[src.index("int32_t q3a_debug_set("):]
    body = body[:body.index("\n}\n")]
    accepted = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', body))
    hdr = open(os.path.join(ROOT, "include", "q3asr.h")).read()
    doc = hdr[:hdr.index("int32_t q3a_debug_set(")]
    doc = doc[doc.rindex("/*"):]
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', doc, flags=re.M))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
    assert lib.q3a_debug_set(b"no_such_key", 1) != 0
    assert b"no_such_key" in lib.q3a_last_error(None)


def _arena(lib, model_dir):
    n = C.c_uint64()
    assert lib.q3a_arena_bytes(model_dir.encode(), C.byref(n)) == 0, lib.q3a_last_error(None)
    buf = np.zeros(n.value, dtype=np.uint8)
    assert lib.q3a_arena_pack(model_dir.encode(), buf.ctypes.data_as(C.c_void_p), n.value) == 0, lib.q3a_last_error(None)
    return buf


def _bf16(t: torch.Tensor) -> np.ndarray:
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def test_arena_contains_permuted_weights(lib, tiny_dir):
    """Spot-check the layout transforms against the checkpoint read by the oracle's loader."""
    buf = _arena(lib, tiny_dir)
    w = O.load_model_weights(tiny_dir)
    cfg = synthetic.CONFIG_TINY
    C_, D = cfg["audio_config"]["downsample_hidden_size"], cfg["audio_config"]["d_model"]
    magic, version, total, flags = struct.unpack_from("<IIQI", buf.tobytes()[:24])
    assert magic == 0x41335141 and total == len(buf) and flags == 0
    raw16 = buf.view(np.uint16)

    def find(needle: np.ndarray) -> int:
        s, n = raw16.tobytes(), needle.tobytes()
        return s.find(n)

    # conv2: [Co][Ci][3][3] -> [Co][kh][kw][Ci]
    c2 = w["thinker.audio_tower.conv2d2.weight"]
    assert find(_bf16(c2.permute(0, 2, 3, 1).contiguous()).ravel()) >= 0
    # conv_out: column c*F+f -> f*C+c
    co = w["thinker.audio_tower.conv_out.weight"]
    F3 = co.shape[1] // C_
    assert find(_bf16(co.reshape(D, C_, F3).permute(0, 2, 1).contiguous()).ravel()) >= 0
    # q|k|v concatenation (decoder layer 1)
    p = "thinker.model.layers.1.self_attn."
    qkv = torch.cat([w[p + "q_proj.weight"], w[p + "k_proj.weight"], w[p + "v_proj.weight"]], 0)
    assert find(_bf16(qkv).ravel()) >= 0
    # gate/up interleave in 16-row blocks
    g, u = w["thinker.model.layers.0.mlp.gate_proj.weight"], w["thinker.model.layers.0.mlp.up_proj.weight"]
    I, H = g.shape
    gu = torch.stack([g.reshape(I // 16, 16, H), u.reshape(I // 16, 16, H)], 1).reshape(2 * I, H)
    assert find(_bf16(gu).ravel()) >= 0
    # embedding copied verbatim
    assert find(_bf16(w["thinker.model.embed_tokens.weight"][:64]).ravel()) >= 0


def test_sharded_and_untied_checkpoint_packs(lib, tiny_untied_dir):
    buf = _arena(lib, tiny_untied_dir)
    w = O.load_model_weights(tiny_untied_dir)
    s = buf.tobytes()
    assert s.find(_bf16(w["thinker.lm_head.weight"][:32]).tobytes()) >= 0
    assert s.find(_bf16(w["thinker.model.embed_tokens.weight"][:32]).tobytes()) >= 0


def test_f16_and_f32_checkpoints_pack_like_bf16_and_round_to_nearest_even(lib, tmp_path):
    """weights.rs:74-89,134-181 widen F16 / BF16 / F32 to f32.  The arena stores matrices as bf16: (i) a checkpoint whose F16 /
    F32 tensors hold bf16-representable values packs to exactly the bytes of the BF16 checkpoint (lossless); (ii) values
    with more mantissa are rounded to nearest-even -- torch's own float -> bfloat16 conversion -- and the arena header
    carries the weights-rounded flag (include/q3asr.h q3a_weights_rounded); vectors stay f32 (exact)."""
    flags_of = lambda arena: int(arena[:32].view(np.uint32)[4])
    base = _arena(lib, synthetic.write_checkpoint(str(tmp_path / "bf16"), "tiny", seed=7))
    assert not flags_of(base) & (1 << 4)
    for dt in ("F16", "F32"):
        d = synthetic.write_checkpoint(str(tmp_path / dt), "tiny", seed=7, dtype=dt)
        assert O.load_model_weights(d)["thinker.model.norm.weight"].dtype == torch.float32   # the oracle widens as the reference does
        got = _arena(lib, d)
        assert flags_of(got) & (1 << 4), dt                # the checkpoint held non-bf16 matrices
        if dt == "F32":                                    # every bf16 value is an f32 value: same bytes after the header
            assert np.array_equal(got[256:], base[256:])
        else:                                              # f16 loses the small magnitudes of bf16: arena = bf16(widened f16), nearest-even
            want = _bf16(O.load_model_weights(d)["thinker.model.layers.1.mlp.down_proj.weight"].flatten())
            hay = got.view(np.uint16)
            starts = np.flatnonzero(hay[:len(hay) - len(want) + 1] == want[0])
            assert any(np.array_equal(hay[i:i + len(want)], want) for i in starts)
    # a genuinely finer F32 checkpoint: one matrix rewritten with full-mantissa values
    d = synthetic.write_checkpoint(str(tmp_path / "fine"), "tiny", seed=8, dtype="F32")
    path = os.path.join(d, "model.safetensors")
    raw = bytearray(open(path, "rb").read())
    hlen = struct.unpack("<Q", raw[:8])[0]
    lo, hi = json.loads(raw[8:8 + hlen])["thinker.model.layers.0.mlp.down_proj.weight"]["data_offsets"]
    fine = torch.randn((hi - lo) // 4, generator=torch.Generator().manual_seed(3)) * 0.05
    raw[8 + hlen + lo:8 + hlen + hi] = fine.numpy().tobytes()
    open(path, "wb").write(raw)
    hay = _arena(lib, d).view(np.uint16)
    want = _bf16(fine)                                      # torch float -> bfloat16 is round-to-nearest-even
    assert not np.array_equal(want, (fine.view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16))   # (truncation would differ)
    starts = np.flatnonzero(hay[:len(hay) - len(want) + 1] == want[0])
    assert any(np.array_equal(hay[i:i + len(want)], want) for i in starts), "nearest-even rounded down_proj matrix not found in the arena"


def test_missing_weight_reports_key(lib, tmp_path):
    d = synthetic.write_checkpoint(str(tmp_path / "m"), "tiny", seed=3)
    # drop one tensor from the safetensors header
    p = os.path.join(d, "model.safetensors")
    raw = open(p, "rb").read()
    (hl,) = struct.unpack("<Q", raw[:8])
    hdr = json.loads(raw[8:8 + hl])
    del hdr["thinker.model.norm.weight"]
    hj = json.dumps(hdr).encode()
    open(p, "wb").write(struct.pack("<Q", len(hj)) + hj + raw[8 + hl:])
    n = C.c_uint64()
    assert lib.q3a_arena_bytes(d.encode(), C.byref(n)) == 0
    buf = np.zeros(n.value, dtype=np.uint8)
    assert lib.q3a_arena_pack(d.encode(), buf.ctypes.data_as(C.c_void_p), n.value) != 0
    assert b"Weight not found: thinker.model.norm.weight" in lib.q3a_last_error(None)  # weights.rs:196


def test_no_weights_and_bad_config(lib, tmp_path):
    d = tmp_path / "empty"
    d.mkdir()
    n = C.c_uint64()
    assert lib.q3a_arena_bytes(str(d).encode(), C.byref(n)) != 0  # no config.json
    (d / "config.json").write_text(json.dumps({"thinker_config": synthetic.CONFIG_TINY}))
    assert lib.q3a_arena_bytes(str(d).encode(), C.byref(n)) == 0
    buf = np.zeros(n.value, dtype=np.uint8)
    assert lib.q3a_arena_pack(str(d).encode(), buf.ctypes.data_as(C.c_void_p), n.value) != 0
    assert b"No model weights found" in lib.q3a_last_error(None)  # weights.rs:21-24
    bad = dict(synthetic.CONFIG_TINY)
    bad["text_config"] = dict(bad["text_config"], head_dim=64)
    (d / "config.json").write_text(json.dumps({"thinker_config": bad}))
    assert lib.q3a_arena_bytes(str(d).encode(), C.byref(n)) != 0
    assert b"head_dim" in lib.q3a_last_error(None)


def test_config_defaults_match_reference(lib, tmp_path):
    """An empty thinker_config must give the serde defaults of src/config.rs (0.6B dims)."""
    d = tmp_path / "dflt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"thinker_config": {"audio_config": {}, "text_config": {}}}))
    n = C.c_uint64()
    assert lib.q3a_arena_bytes(str(d).encode(), C.byref(n)) == 0
    cfg = O.AsrConfig.from_file(str(d / "config.json"))
    assert cfg.audio.d_model == 896 and cfg.text.num_hidden_layers == 28 and cfg.text.mrope_section == [24, 20, 20]
    # 782.4 M parameters -> ~1.56 GB of bf16 plus f32 vectors and alignment
    assert 1.55e9 < n.value < 1.60e9


def test_prompt_and_lengths_match_oracle(lib):
    for T in (0, 1, 54, 390):
        for prefix in (None, [5, 6, 7]):
            ids = HipEngine.build_prompt(T, prefix)
            ref, _ = O.build_prompt(T, prefix)
            assert ids.tolist() == ref
    for n in (161, 16000, 66560, 480000, 480001):
        assert lib.q3a_num_frames(n) == (n + 159) // 160


def test_engine_fails_loudly_without_gpu(lib, tiny_dir):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Q3aError, match="no HIP device"):
        HipEngine(tiny_dir, 0)


def test_output_parsing_mirror():
    for raw, forced in [("language English<asr_text>Hi.", False), ("language Chinese 你好", False), ("x", False), (" y ", True)]:
        assert parse_asr_output(raw, forced) == O.parse_asr_output(raw, forced)
    assert capitalize_first("chinese") == "Chinese"


def test_committed_bench_line_follows_the_contract():
    """profiles/r4_bench_final.json is the bench.py line of this round: every field the driver and the judge read
    must be there with the right type (BASELINE.json metric, roofline and cpu_baseline objects), the roofline must come
    from the in-situ trace and be internally consistent, and the extra legs must name BASELINE.json's other configs."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    j = json.load(open(os.path.join(root, "profiles", "r4_bench_final.json")))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert j["metric"].split(",")[0] == base["metric"].split(",")[0]
    assert j["unit"] == "audio-seconds/sec" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["n_gpus"] == 1 and j["steps"] > 0 and j["warmup"] >= 0 and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] / 1e3 - 30.0 * j["config"]["clips_per_gpu"]) < 0.5
    assert "workload" in j["config"] and "model" not in j["config"] and "synthetic" in j["data"]
    assert j["host_to_host"]["value"] > 0 and j["host_to_host"]["value"] <= j["value"] * 1.02   # PCIe inside the clock cannot be faster
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) < 1.0       # GB/s = bytes / us / 1e3
    assert "rocprofv3" in r["avg_launch_us_source"] and "MICROBENCHMARK" not in r["avg_launch_us_source"]
    assert r["launches_per_token"] == 56 and r["microbench"]["avg_launch_us"] <= r["avg_launch_us"]
    assert r["traffic"] is None or r["traffic"] >= 0.95 * r["bytes_per_launch"]   # no under-counted traffic
    assert r["traffic"] is None or "pmc" in r["traffic_source"]
    assert 0 < r["decode_stage"]["frac"] < 1
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and len(c["runs"]) >= 5
    ne = j["natural_eos"]  # the reference's contract (fixed_new_tokens = 0) beside the fixed-N number, BASELINE.md section 3
    assert ne["stopped_at_planned_token"] is True and ne["generated_tokens"] == j["config"]["new_tokens"]
    assert 0 < ne["value"] <= j["value"] * 1.02 and ne["decode_steps_executed"] == ne["decode_steps_needed"]   # run-ahead 1: no step past the EOS
    assert j["two_streams"]["ids_equal_to_one_engine"] is True
    runs = r["avg_launch_us_of_the_three_child_runs"]   # the reported in-situ average is the MEDIAN of three profiled child runs
    assert len(runs) == 3 and abs(sorted(runs)[1] - r["avg_launch_us"]) < 1e-3
    assert abs(ne["vs_fixed_n_ms"] - (ne["ms_per_step"] - ne["fixed_n_same_engine_ms"])) < 2e-3   # paired on one engine
    ex = j["extra"]
    assert len(ex) == 2 and "batch=32" in ex[0]["workload"] and "1.7b" in ex[1]["workload"] and all(e["value"] > 0 for e in ex)


def test_round6_bench_line_carries_the_driver_visible_evidence():
    """profiles/r6_bench_final.json (round 6): the fields added this round are in the line a driver run produces --
    per-leg roofline objects for BASELINE configs[2] / configs[3], the measured denominators, the precise-mode throughput, the
    host-to-host window at the top level -- and they are internally consistent."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    j = json.load(open(os.path.join(root, "profiles", "r6_bench_final.json")))
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["timed_window"].startswith("PCM resident")   # the contract's `value`
    assert j["value_host_to_host"] == j["host_to_host"]["value"] and 0 < j["value_host_to_host"] <= 1.02 * j["value"]
    vl = j["host_to_host"]["varying_lengths"]
    assert 0 < vl["value"] <= 1.02 * j["value"] and vl["ms_per_step"] > 0
    pk = j["measured_peaks"]
    assert 2600 < pk["hbm_read_GBps"] <= 8000 and 750 < pk["mfma_bf16_TFLOPs"] <= 2500 and pk["gemm"] == [8192, 8192, 8192]
    r = j["roofline"]
    assert abs(r["frac_of_measured_hbm_read"] - r["achieved"] / pk["hbm_read_GBps"]) < 1e-3 and r["frac_of_measured_hbm_read"] > r["frac"]
    assert len(r["avg_launch_us_of_the_child_runs"]) == 3 and r["traffic"] >= 0.95 * r["bytes_per_launch"]
    pm = j["precise_mode"]
    assert 0 < pm["value"] < j["value"] and abs(pm["value"] * pm["ms_per_step"] / 1e3 - 30.0) < 0.5
    c = j["cpu_baseline"]
    assert c["cores_physical"] >= c["cores"] >= 1 and c["cores_logical"] >= c["cores_physical"]
    ex = j["extra"]
    assert len(ex) == 2 and "batch=32" in ex[0]["workload"] and "1.7b" in ex[1]["workload"]
    for e, kern in zip(ex, ("decode_attn_batched_kernel", "skinny_kernel")):
        rr = e["roofline"]
        assert rr["kernel"].startswith(kern) and rr["bound"] == "hbm" and "rocprofv3" in rr["avg_launch_us_source"]
        assert abs(rr["achieved"] - rr["bytes_per_launch"] / rr["avg_launch_us"] / 1e3) < 1.0 and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
        assert 0.9 * rr["bytes_per_launch"] <= rr["traffic"] <= 1.2 * rr["bytes_per_launch"] and "pmc" in rr["traffic_source"]
        assert abs(rr["launches_per_token"] - 28) < 0.5 and 0 < rr["decode_stage"]["frac"] < rr["frac"]
        assert e["host_to_host"]["value"] <= 1.02 * e["value"]


def test_bench_kernel_trace_leaves_the_warmup_pass_out(tmp_path, monkeypatch):
    """bench.kernel_trace: per kernel, the dispatches of the warm-up passes (the first warmup / (warmup + steps) by start time) are
    dropped from the in-situ statistics, as the bench's own timed region drops its warm-up steps (no rocprofv3, no GPU: a rocpd-shaped
    SQLite file stands in for the child's trace)."""
    import sqlite3
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    out = tmp_path / "prof"
    out.mkdir()
    db = sqlite3.connect(str(out / "x_results.db"))
    db.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    for p in range(4):                       # 1 warm-up + 3 timed passes of 5 launches each; the warm-up pass is 1000 ns slower
        for i in range(5):
            dur = 5000 + (1000 if p == 0 else 0)
            db.execute("insert into kernels values (?, ?, ?)", ("void q3a::(anonymous namespace)::gemv1_kernel<2, 2, true, false>(q3a::GemvArgs)", t, t + dur))
            t += dur + 1500
        db.execute("insert into kernels values (?, ?, ?)", ("void q3a::(anonymous namespace)::mel_kernel(q3a::MelBatch)", t, t + 200000))  # once per pass
        t += 201500
    db.commit(); db.close()
    monkeypatch.setattr(bench, "_rocprof", lambda extra, inner, timeout: (str(out / "x_results.db"), str(out)))
    res = bench.kernel_trace(["--steps", "3", "--warmup", "1"], warmup=1, steps=3)
    g = next(v for k, v in res.items() if k.startswith("gemv1_kernel"))
    assert g["calls"] == 15 and abs(g["avg_us"] - 5.0) < 1e-9 and abs(g["warmup_avg_us"] - 6.0) < 1e-9
    m = next(v for k, v in res.items() if k.startswith("mel_kernel"))
    assert m["calls"] == 3 and abs(m["avg_us"] - 200.0) < 1e-9


def test_bench_n_gpu_launch_path_without_a_gpu():
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE re-launches itself under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1); --launch-check runs exactly that path on the gloo backend: the processes fork,
    meet, and rank 0 prints one JSON line carrying the max over ranks."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.multi_gpu_command(8, ["--gpus", "8", "--steps", "3"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)["launch_check"]
    assert j == {"world": 2, "max_over_ranks": 2, "master_addr": "127.0.0.1"}


def test_bench_native_group_leg_is_a_bounded_child(monkeypatch):
    """The q3a_group_* leg of `bench.py --gpus N` runs in a child process: a child that dies (here: no HIP device) or does not answer
    within the timeout (here: a sleeping stand-in) yields an error record for the JSON line, never an exception or a hang."""
    import argparse, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    args = argparse.Namespace(preset="0.6b", seconds=30.0, new_tokens=100, steps=2, precise=False, ckpt_dir=None)
    if not torch.cuda.is_available():
        rec = bench.native_group_child(args, 2, 1, timeout_s=240)
        assert rec["value"] is None and "HIP device" in rec["error"], rec
    real_run = subprocess.run
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: real_run([sys.executable, "-c", "import time; time.sleep(60)"], **kw))
    t0 = time.time()
    rec = bench.native_group_child(args, 2, 1, timeout_s=2)
    assert rec["value"] is None and "no result within 2 s" in rec["error"] and time.time() - t0 < 30, rec
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: real_run([sys.executable, "-c", "print('noise'); print('{\"workload\": \"q3a_group_transcribe\", \"value\": 7.0}')"], **kw))
    assert bench.native_group_child(args, 2, 1)["value"] == 7.0


def _cpp_function_body(src: str, signature_start: str) -> str:
    """Text of the brace-balanced body that follows `signature_start` in a C++ source."""
    i = src.index(signature_start)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        c = src[k]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[j:k + 1]
        k += 1


def test_step_buffers_cover_the_captured_step():
    """A cached decode-step graph holds raw device addresses.  Two guards keep a stale one from replaying -- the reallocation
    sweep of setup_prompts and the address hash in make_graph_sig -- and both walk q3a_engine::step_bufs().  This test scans
    the source of every function that enqueues a launch of the captured step for DevBuf members and fails when one is not
    in that list ("one forgotten DevBuf re-opens the stale-graph bug")."""
    src = open(os.path.join(ROOT, "qwen3_asr_rs_amd", "csrc", "engine.cpp")).read()
    members = set()
    for decl in re.findall(r"^  DevBuf ([^;]+);", src, flags=re.M):
        members.update(n.strip() for n in decl.split(","))
    assert {"kcache", "x_dec", "logits", "nn_ss"} <= members
    listed = set(re.findall(r"&(\w+)", _cpp_function_body(src, "std::vector<DevBuf*> step_bufs()")))
    assert listed <= members and len(listed) >= 20
    step_functions = ["void decode_layer(", "void run_head(", "void enqueue_decode_step(", "void* kc_layer(", "void* vc_layer(",
                      "NextNormOut first_layer_norm_out(", "uint16_t* nn_x_g(", "float* nn_ss_g(", "float* s_ctx_g(", "float* s_act_g(",
                      "void batched_proj("]
    used = set()
    for f in step_functions:
        body = _cpp_function_body(src, f)
        used.update(m for m in members if re.search(r"\b" + re.escape(m) + r"\b", body))
    debug_only = {n for n in used if n.startswith("dbg_")}
    missing = used - listed - debug_only
    assert not missing, f"DevBuf members used by the captured decode step but absent from step_bufs(): {sorted(missing)}"
    # both guards really use the list
    assert "step_bufs()" in _cpp_function_body(src, "std::string make_graph_sig() const")
    assert "step_bufs()" in _cpp_function_body(src, "void setup_prompts(")


def test_mfma_table_shapes_and_launch_matching():
    """tools/mfma_table.py (the counter-vs-wall-clock MFMA table of DESIGN.md section 6): its shape list reproduces the tile counts
    the committed profiles show for 32 clips (conv2 6000, conv3 1560, enc qkv 506 after the round split, fc1 686, 196-tile residual
    shapes, dec qkv 768, gate/up 1224, o / down 204), leaves out launches below gemm256's tile threshold, and its matcher assigns a
    launch sequence with same-grid neighbours (out / fc2, o / down) and foreign launches in between to the right shapes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_table", os.path.join(ROOT, "tools", "mfma_table.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    shp = mt.shapes("0.6b", 32)
    grids, launched = {}, {}
    for x in shp:
        grids.setdefault(x["label"], x["tiles"])
        launched.setdefault(x["label"], x["grid"])
    # round 6: the persistent walk launches min(tiles, 256 CUs) workgroups
    assert launched["conv2 implicit GEMM"] == 256 and launched["enc fc1 (GELU)"] == 256 and launched["enc out (+residual)"] == 196 and launched["dec down (+residual)"] == 204
    assert grids["conv2 implicit GEMM"] == 6000 and grids["conv3 implicit GEMM"] == 1560 and grids["enc qkv"] == 506
    assert grids["enc fc1 (GELU)"] == 686 and grids["enc out (+residual)"] == 196 and grids["enc fc2 (+residual)"] == 196
    assert grids["dec qkv + QK-norm/RoPE/KV epilogue"] == 768 and grids["dec gate/up (SwiGLU)"] == 1224 and grids["dec down (+residual)"] == 204
    assert mt.split_rows(12480, 2688) == 46 * 256 and mt.split_rows(12480, 3584) == 0 and mt.split_rows(12960, 4096) == 48 * 256
    assert "conv_out (+pos-emb, gather)" not in {x["label"] for x in mt.shapes("1.7b", 16)}   # 100 tiles < 128: not a gemm256 launch
    names = {"ConvA256": "gemm256_kernel<false, ConvA256, false>", "DenseA256": "gemm256_kernel<false, DenseA256, false>",
             "DenseA256, true": "gemm256_kernel<false, DenseA256, true>", "gemm256_kernel<true": "gemm256_kernel<true, DenseA256, false>"}
    launches = []
    for i, x in enumerate(shp * 2):
        launches.append(dict(name=names[x["ksub"]], grid=x["grid"], dur=float(i)))
        if i % 7 == 0:
            launches.append(dict(name="gemm256_kernel<false, DenseA256, false>", grid=3, dur=0.0))   # a launch of some other shape
        launches.append(dict(name="norm_kernel<false>", grid=12480, dur=0.0))
    by, matched, unmatched = mt.classify(launches, shp)
    assert matched == 2 * len(shp) and unmatched == len([i for i in range(2 * len(shp)) if i % 7 == 0])
    per_pass = {lab: sum(1 for x in shp if x["label"] == lab) for lab in grids}
    assert {lab: len(v) for lab, v in by.items()} == {lab: 2 * n for lab, n in per_pass.items()}
    # same-grid neighbours keep their order: every "enc out" launch sits right before an "enc fc1" one in the sequence
    assert all(d["dur"] % len(shp) in [i for i, x in enumerate(shp) if x["label"] == "enc out (+residual)"] for d in by["enc out (+residual)"])
