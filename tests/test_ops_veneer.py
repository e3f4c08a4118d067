"""Op-level veneer (include/q3asr_ops.h): every `q3a_op_*` entry point against the torch-CPU op the
reference's tch arm forwards to (src/tensor.rs:145-488, operators :960-1161), then the reference's own op SEQUENCES --
WhisperFeatureExtractor::extract (src/mel.rs:49-96), AudioEncoderLayer::forward (src/layers.rs:152-172,192-195,230-242) and
TextDecoderLayer::forward (src/layers.rs:48-54,284-375,396-400,442-463) -- written against the Python mirror of `struct
Tensor` (qwen3_asr_rs_amd/tensor.py: same method names) and compared with the oracle.  Tolerance: fp32 rounding (different
summation orders), stated per check.
"""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd import tensor as T
from qwen3_asr_rs_amd.tensor import Tensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------
# CPU: the library exports every symbol the header declares; host arrays behave; no CPU compute path
# ---------------------------------------------------------------------------------------------------------------
def test_ops_exports_match_header(lib):
    syms = T.declared_symbols()
    assert len(syms) >= 70
    for s in syms:
        assert hasattr(lib, s), f"libq3asr_hip.so does not export {s}"
    T.lib()  # argtypes parse for every declaration


def test_every_must_have_method_of_the_tch_arm_has_an_entry_point():
    """the methods L3/L4 actually call.  Each must exist as q3a_op_<name> / q3a_array_<name>."""
    must = ("from_slice_f32 from_slice_i64 zeros full cat embedding hann_window reshape narrow unsqueeze squeeze_dim transpose "
            "permute expand contiguous tr get matmul clamp_min maximum abs square rsqrt log10 softmax gelu silu mean_dim max argmax "
            "triu slice_scatter layer_norm conv2d reflection_pad1d stft to_dtype to_device shallow_clone add sub mul div "
            "add_scalar sub_scalar mul_scalar div_scalar neg add_inplace").split()
    syms = set(T.declared_symbols())
    for m in must:
        assert f"q3a_op_{m}" in syms, m
    for m in ("shape", "ndim", "dtype", "device", "int64_value", "f64_value", "to_vec_f32", "free"):
        assert f"q3a_array_{m}" in syms, m
    # and the Rust shim binds exactly these names
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "backend", "hip", "ffi.rs")).read()
    bound = set(re.findall(r"pub fn (q3a_[a-z0-9_]+)\s*\(", ffi))
    assert bound == syms, (sorted(syms - bound), sorted(bound - syms))


def test_host_arrays_and_errors_without_gpu():
    t = Tensor.from_slice_f32([1, 2, 3, 4, 5, 6])
    assert t.size() == [6] and t.kind() == T.F32 and t.device() == T.CPU and t.dim() == 1
    r = t.reshape([2, -1])
    assert r.size() == [2, 3] and r.transpose(0, 1).size() == [3, 2] and r.f64_value([1, 2]) == 6.0
    assert r.transpose(0, 1).contiguous().to_vec_f32().tolist() == [1, 4, 2, 5, 3, 6]
    assert r.narrow(1, 1, 2).to_vec_f32().tolist() == [2, 3, 5, 6]
    i = Tensor.from_slice_i64([7, 8, 9])
    assert i.kind() == T.I64 and i.int64_value([2]) == 9 and i.to_dtype(T.F32).to_vec_f32().tolist() == [7, 8, 9]
    assert i.unsqueeze(0).size() == [1, 3] and i.unsqueeze(0).squeeze_dim(0).size() == [3]
    with pytest.raises(T.OpsError, match="host array"):
        t.gelu()
    with pytest.raises(T.OpsError, match="size mismatch"):
        t.reshape([4, 2])
    with pytest.raises(T.OpsError, match="out of range"):
        t.narrow(3, 0, 1)
    if not torch.cuda.is_available():
        with pytest.raises(T.OpsError, match="no HIP device"):
            Tensor.zeros([2, 2], T.F32, 0)


# ---------------------------------------------------------------------------------------------------------------
# GPU: op by op against torch CPU
# ---------------------------------------------------------------------------------------------------------------
def dev(x):
    x = np.asarray(x)
    if x.dtype == np.float64:
        x = x.astype(np.float32)
    return Tensor.from_numpy(x, 0)


def close(t: Tensor, ref, atol=1e-6, rtol=1e-5):
    ref = ref.detach().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert t.size() == list(ref.shape), (t.size(), ref.shape)
    np.testing.assert_allclose(t.numpy(), ref.astype(np.float32), atol=atol, rtol=rtol)


@pytest.mark.gpu
def test_ops_against_torch():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(3, 4, 5, generator=g)
    b = torch.randn(4, 5, generator=g)
    A, B = dev(a.numpy()), dev(b.numpy())
    # creation
    close(Tensor.zeros([2, 3]), torch.zeros(2, 3)); close(Tensor.ones([2, 3]), torch.ones(2, 3)); close(Tensor.full([2], 2.5), torch.full((2,), 2.5))
    assert Tensor.arange(3, 8).to_dtype(T.F32).to_vec_f32().tolist() == [3, 4, 5, 6, 7]
    close(Tensor.arange_f(0.0, 1.0, 0.25), torch.arange(0.0, 1.0, 0.25))
    close(Tensor.hann_window(400), torch.hann_window(400), atol=1e-6)
    close(Tensor.from_slice_f32(a.flatten().numpy()).to_dtype(T.F32).to_device(0).reshape([3, 4, 5]), a)
    # views
    close(A.transpose(0, 2), a.transpose(0, 2)); close(A.permute([2, 0, 1]), a.permute(2, 0, 1)); close(A.narrow(1, 1, 2), a.narrow(1, 1, 2))
    close(A.narrow(-1, 2, 3), a.narrow(-1, 2, 3)); close(A.unsqueeze(1), a.unsqueeze(1)); close(A.unsqueeze(-1), a.unsqueeze(-1))
    close(A.get(1), a[1]); close(A.select(2, 3), a.select(2, 3)); close(B.tr(), b.t()); close(A.transpose(0, 1).reshape([4, 15]), a.transpose(0, 1).reshape(4, 15))
    close(A.permute([1, 0, 2]).contiguous().view([4, -1]), a.permute(1, 0, 2).contiguous().view(4, -1))
    close(B.unsqueeze(0).expand([3, -1, -1]), b.unsqueeze(0).expand(3, -1, -1)); close(A.narrow(0, 0, 1).squeeze_dim(0), a[:1].squeeze(0))
    close(Tensor.cat([A, A.narrow(1, 0, 2)], 1), torch.cat([a, a[:, :2]], 1)); close(Tensor.cat([A.transpose(0, 1), A.transpose(0, 1)], -1), torch.cat([a.transpose(0, 1)] * 2, -1))
    close(Tensor.stack([B, B], 0), torch.stack([b, b], 0)); close(A.shallow_clone(), a)
    # arithmetic with broadcasting, scalars, in-place
    close(A + B, a + b); close(A - B, a - b); close(A * B, a * b); close(A / B, a / b); close(A.maximum(B), torch.maximum(a, b))
    close(A + 4.0, a + 4.0); close(A - 8.0, a - 8.0); close(A * 0.5, a * 0.5); close(A / 4.0, a / 4.0); close(-A, -a); close(A.neg(), -a)
    close(A.transpose(0, 2) * A.transpose(0, 2), (a * a).transpose(0, 2)); close(A.narrow(2, 0, 1) + B.unsqueeze(0), a[:, :, :1] + b.unsqueeze(0))
    C = A + 0.0; C += B; close(C, a + b); close(A, a)   # (+= writes through: a fresh array keeps A intact)
    Z = Tensor.zeros([2, 2]); Z.fill_(3.0); close(Z, torch.full((2, 2), 3.0))
    # math
    p = a.abs() + 0.1
    P = dev(p.numpy())
    close(A.abs(), a.abs()); close(A.square(), a.square()); close(P.sqrt(), p.sqrt()); close(P.rsqrt(), p.sqrt().reciprocal(), rtol=2e-6)
    close(P.log10(), p.log10(), atol=2e-6); close(A.sin(), a.sin(), atol=2e-6); close(A.cos(), a.cos(), atol=2e-6); close(A.exp(), a.exp(), rtol=2e-6)
    close(A.clamp_min(0.2), a.clamp_min(0.2)); close(P.pow_scalar(1.5), p.pow(1.5), rtol=1e-5)
    close(A.gelu(), F.gelu(a), atol=2e-6); close(A.silu(), F.silu(a), atol=2e-6)
    close(A.softmax(-1), a.softmax(-1), atol=2e-7); close(A.softmax(1), a.softmax(1), atol=2e-7)
    close(A.mean_dim([-1], True), a.mean(-1, keepdim=True), atol=1e-6); close(A.mean_dim([0, 2], False), a.mean((0, 2)), atol=1e-6)
    assert abs(A.max().f64_value([]) - float(a.max())) == 0.0
    assert A.argmax(-1, False).to_dtype(T.F32).numpy().tolist() == a.argmax(-1).tolist()
    assert A.argmax(1, True).size() == [3, 1, 5] and A.argmax(1, True).to_dtype(T.F32).numpy().tolist() == a.argmax(1, keepdim=True).tolist()
    tie = dev(np.array([[1.0, 3.0, 3.0, 2.0]], dtype=np.float32))
    assert tie.argmax(-1, False).int64_value([0]) == 1                       # first index on ties (tensor.rs:370-372)
    close(Tensor.full([1, 1, 4, 6], float("-inf")).triu(3), torch.full((1, 1, 4, 6), float("-inf")).triu(3))
    close(A.triu(-1), a.triu(-1))
    src = torch.randn(3, 2, 5, generator=g)
    close(A.slice_scatter(dev(src.numpy()), 1, 1, 3, 1), a.slice_scatter(src, 1, 1, 3, 1))
    close(A.slice_scatter(dev(src.numpy()), 1, 0, 4, 2), a.slice_scatter(src, 1, 0, 4, 2))
    w, bias = torch.randn(5, generator=g), torch.randn(5, generator=g)
    close(A.layer_norm([5], dev(w.numpy()), dev(bias.numpy()), 1e-5), F.layer_norm(a, (5,), w, bias, 1e-5), atol=2e-6)
    close(A.layer_norm([5], None, None, 1e-5), F.layer_norm(a, (5,), None, None, 1e-5), atol=2e-6)
    a8, w8, b8 = torch.randn(6, 8, generator=g), torch.randn(8, generator=g), torch.randn(8, generator=g)   # D % 4 == 0: the engine's kernel
    close(dev(a8.numpy()).layer_norm([8], dev(w8.numpy()), dev(b8.numpy()), 1e-5), F.layer_norm(a8, (8,), w8, b8, 1e-5), atol=2e-6)
    # matmul: 2-D, batched x 2-D weight.tr() (Linear), batched x batched, broadcast batch, vectors
    m1, m2 = torch.randn(37, 70, generator=g), torch.randn(70, 45, generator=g)
    close(dev(m1.numpy()).matmul(dev(m2.numpy())), m1 @ m2, atol=2e-5)
    wt = torch.randn(33, 70, generator=g)
    x3 = torch.randn(2, 9, 70, generator=g)
    close(dev(x3.numpy()).matmul(dev(wt.numpy()).tr()), x3 @ wt.t(), atol=2e-5)
    q, k = torch.randn(2, 3, 7, 16, generator=g), torch.randn(2, 3, 11, 16, generator=g)
    close(dev(q.numpy()).matmul(dev(k.numpy()).transpose(-2, -1)), q @ k.transpose(-2, -1), atol=1e-5)
    close(dev(q.numpy()).matmul(dev(k.numpy()[0:1, 0:1]).transpose(-2, -1)), q @ k[0:1, 0:1].transpose(-2, -1), atol=1e-5)
    close(dev(m1.numpy()).matmul(dev(m2.numpy()[:, 0])), m1 @ m2[:, 0], atol=2e-5)
    close(dev(m1.numpy()[0]).matmul(dev(m2.numpy())), m1[0] @ m2, atol=2e-5)
    # embedding, conv2d (the three convolutions of the stem at small size), reflection pad, stft
    emb = torch.randn(50, 8, generator=g)
    idx = torch.tensor([[3, 49, 0], [7, 7, 1]])
    close(Tensor.embedding(dev(emb.numpy()), dev(idx.numpy())), F.embedding(idx, emb))
    xin = torch.randn(2, 1, 16, 10, generator=g)
    w1, b1 = torch.randn(6, 1, 3, 3, generator=g), torch.randn(6, generator=g)
    w2 = torch.randn(4, 6, 3, 3, generator=g)
    c1 = dev(xin.numpy()).conv2d(dev(w1.numpy()), dev(b1.numpy()), [2, 2], [1, 1], [1, 1], 1)
    close(c1, F.conv2d(xin, w1, b1, 2, 1), atol=1e-5)
    close(c1.conv2d(dev(w2.numpy()), None, [2, 2], [1, 1], [1, 1], 1), F.conv2d(F.conv2d(xin, w1, b1, 2, 1), w2, None, 2, 1), atol=2e-5)
    sig = torch.randn(1, 1, 1000, generator=g)
    close(dev(sig.numpy()).reflection_pad1d([200, 200]), F.pad(sig, (200, 200), mode="reflect"))
    wav = F.pad(sig, (200, 200), mode="reflect")[0, 0]
    win = torch.hann_window(400)
    ref = torch.stft(wav, 400, 160, 400, win, center=False, normalized=False, onesided=True, return_complex=True)
    S = dev(wav.numpy()).stft(400, 160, 400, Tensor.hann_window(400), False, True, True)
    assert S.kind() == T.C64 and S.size() == list(ref.shape)
    close(S.abs(), ref.abs(), atol=2e-4, rtol=1e-4)
    close(dev(wav.numpy()).stft(400, 160, 400, Tensor.hann_window(400), False, True, False), torch.view_as_real(ref), atol=2e-4, rtol=1e-4)
    # dtype / device round trips
    close(A.to_dtype(T.BF16).to_dtype(T.F32), a.to(torch.bfloat16).float()); close(A.to_dtype(T.F16).to_dtype(T.F32), a.half().float())
    assert A.to_dtype(T.I64).to_dtype(T.F32).numpy().tolist() == a.to(torch.int64).float().tolist()
    assert A.to_device(T.CPU).device() == T.CPU and A.to_device(T.CPU).to_vec_f32().tolist() == a.flatten().tolist()
    assert A.kind() == T.F32 and A.device() == 0
    with pytest.raises(T.OpsError, match="broadcast"):
        A + dev(np.zeros((3, 3), np.float32))


# ---------------------------------------------------------------------------------------------------------------
# GPU: the reference's op sequences over the veneer vs the oracle
# ---------------------------------------------------------------------------------------------------------------
def extract(samples, mel_filters: Tensor, n_fft=400, hop=160) -> Tensor:
    """src/mel.rs:49-96, call for call."""
    padded_len = (len(samples) + hop - 1) // hop * hop
    padded = np.zeros(padded_len, np.float32)
    padded[:len(samples)] = samples
    waveform = Tensor.from_slice_f32(padded).to_dtype(T.F32).to_device(0)
    window = Tensor.hann_window(n_fft, 0)
    pad = n_fft // 2
    waveform = waveform.unsqueeze(0).unsqueeze(0)
    waveform = waveform.reflection_pad1d([pad, pad]).squeeze_dim(0).squeeze_dim(0)
    stft = waveform.stft(n_fft, hop, n_fft, window, False, True, True)
    magnitudes = stft.abs().square()
    num_frames = magnitudes.size()[1]
    magnitudes = magnitudes.narrow(1, 0, num_frames - 1)
    mel_spec = mel_filters.matmul(magnitudes)
    log_mel = mel_spec.clamp_min(1e-10).log10()
    max_val = log_mel.max()
    log_mel = log_mel.maximum(max_val - 8.0)
    return (log_mel + 4.0) / 4.0


class W:
    """weights.rs get_weight over the oracle's fp32 weight dict, moved to the device through q3a_op_from_bytes."""

    def __init__(self, weights):
        self.w, self.cache = weights, {}

    def __call__(self, key) -> Tensor:
        if key not in self.cache:
            self.cache[key] = Tensor.from_numpy(self.w[key].numpy(), 0)
        return self.cache[key]


def linear(wts, prefix, x, bias=True):          # layers.rs:74-80
    out = x.matmul(wts(prefix + ".weight").tr())
    return out + wts(prefix + ".bias") if bias and (prefix + ".bias") in wts.w else out


def layer_norm(wts, prefix, x):                   # layers.rs:25-28
    return x.layer_norm([x.size()[-1]], wts(prefix + ".weight"), wts(prefix + ".bias"), 1e-5)


def rms_norm(weight: Tensor, x, eps):             # layers.rs:48-54
    x = x.to_dtype(T.F32)
    variance = (x * x).mean_dim([-1], True)
    x = x * (variance + eps).rsqrt()
    return (x * weight).to_dtype(T.F32)


def audio_encoder_layer(wts, p, x, mask, nh):     # layers.rs:152-172, 192-195, 230-242
    residual = x
    h = layer_norm(wts, p + ".self_attn_layer_norm", x)
    bsz, seq_len, d = h.size()
    hd = d // nh
    q = linear(wts, p + ".self_attn.q_proj", h).reshape([bsz, seq_len, nh, hd]).permute([0, 2, 1, 3])
    k = linear(wts, p + ".self_attn.k_proj", h).reshape([bsz, seq_len, nh, hd]).permute([0, 2, 1, 3])
    v = linear(wts, p + ".self_attn.v_proj", h).reshape([bsz, seq_len, nh, hd]).permute([0, 2, 1, 3])
    attn = q.matmul(k.transpose(-2, -1)) / float(np.sqrt(hd))
    if mask is not None:
        attn = attn + mask
    attn = attn.softmax(-1).to_dtype(T.F32)
    out = attn.matmul(v).permute([0, 2, 1, 3]).reshape([bsz, seq_len, nh * hd])
    x = linear(wts, p + ".self_attn.out_proj", out) + residual
    residual = x.shallow_clone()
    h = layer_norm(wts, p + ".final_layer_norm", x)
    h = linear(wts, p + ".fc2", linear(wts, p + ".fc1", h).gelu())
    return h + residual


def rotate_half(x):                               # layers.rs:370-375
    half = x.size()[-1] // 2
    return Tensor.cat([-x.narrow(-1, half, half), x.narrow(-1, 0, half)], -1)


def apply_rotary_emb(x, cos, sin):                # layers.rs:361-367
    cos, sin = cos.unsqueeze(0).unsqueeze(0), sin.unsqueeze(0).unsqueeze(0)
    return x * cos + rotate_half(x) * sin


def repeat_kv(x, n_rep):                          # layers.rs:350-358
    if n_rep == 1:
        return x.shallow_clone()
    bsz, nkv, seq_len, hd = x.size()
    return x.unsqueeze(2).expand([bsz, nkv, n_rep, seq_len, hd], False).reshape([bsz, nkv * n_rep, seq_len, hd])


def text_decoder_layer(wts, p, x, cos, sin, mask, tc):   # layers.rs:284-342, 396-400, 442-463
    residual = x
    h = rms_norm(wts(p + ".input_layernorm.weight"), x, tc.rms_norm_eps)
    bsz, seq_len, _ = h.size()
    nq, nkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    q = linear(wts, p + ".self_attn.q_proj", h).reshape([bsz, seq_len, nq, hd]).transpose(1, 2)
    k = linear(wts, p + ".self_attn.k_proj", h).reshape([bsz, seq_len, nkv, hd]).transpose(1, 2)
    v = linear(wts, p + ".self_attn.v_proj", h).reshape([bsz, seq_len, nkv, hd]).transpose(1, 2)
    q = rms_norm(wts(p + ".self_attn.q_norm.weight"), q, tc.rms_norm_eps)
    k = rms_norm(wts(p + ".self_attn.k_norm.weight"), k, tc.rms_norm_eps)
    q, k = apply_rotary_emb(q, cos, sin), apply_rotary_emb(k, cos, sin)
    k, v = repeat_kv(k, nq // nkv), repeat_kv(v, nq // nkv)
    attn = q.matmul(k.transpose(-2, -1)) / float(np.sqrt(hd))
    if mask is not None:
        attn = attn + mask
    attn = attn.softmax(-1).to_dtype(T.F32)
    out = attn.matmul(v).transpose(1, 2).reshape([bsz, seq_len, nq * hd])
    x = linear(wts, p + ".self_attn.o_proj", out) + residual
    residual = x.shallow_clone()
    h = rms_norm(wts(p + ".post_attention_layernorm.weight"), x, tc.rms_norm_eps)
    gate = linear(wts, p + ".mlp.gate_proj", h).silu()
    up = linear(wts, p + ".mlp.up_proj", h)
    return linear(wts, p + ".mlp.down_proj", gate * up) + residual


@pytest.mark.gpu
def test_reference_op_sequences_over_the_veneer_match_the_oracle(tiny_dir):
    from oracle import q3asr_oracle as O
    orc = O.AsrOracle(tiny_dir)
    clip = synthetic.synthetic_clip(0, 9.3)   # 10 chunks -> the dense window mask is exercised
    r = orc.transcribe_ids(clip, fixed_new_tokens=1, want_taps=True)
    # mel.rs over the veneer
    filt = Tensor.from_numpy(O.create_mel_filterbank().numpy(), 0)
    mel = extract(clip, filt)
    assert mel.size() == list(r.taps["mel"].shape)
    assert float(np.abs(mel.numpy() - r.taps["mel"].numpy()).max()) <= 1e-4
    # encoder layer 0 (with the reference's dense -inf window mask) : enc_in -> enc_layer0
    wts = W(orc.weights)
    ac = orc.cfg.audio
    x = Tensor.from_numpy(r.taps["enc_in"].numpy()[None], 0)
    chunk_tokens = [13] * 9 + [O.feat_extract_output_length(30)]
    mask = Tensor.from_numpy(O.build_window_mask(r.num_audio_tokens, chunk_tokens, ac.n_window, ac.n_window_infer).numpy(), 0)
    y = audio_encoder_layer(wts, "thinker.audio_tower.layers.0", x, mask, ac.encoder_attention_heads)
    ref = r.taps["enc_layer0"].numpy()
    assert float(np.linalg.norm(y.numpy()[0] - ref) / np.linalg.norm(ref)) <= 1e-5
    # decoder layer 0 on the prefill: dec_embed -> dec_layer0, causal mask built as text_decoder.rs:121-131 does
    tc = orc.cfg.text
    P = r.prompt_len
    cos, sin = O.compute_mrope_cos_sin([list(range(P))] * 3, tc.head_dim, tc.rope_theta, tc.mrope_section, tc.mrope_interleaved)
    causal = Tensor.full([1, 1, P, P], float("-inf"), T.F32, 0).triu(1)
    h = Tensor.from_numpy(r.taps["dec_embed"].numpy()[None], 0)
    y = text_decoder_layer(wts, "thinker.model.layers.0", h, Tensor.from_numpy(cos.numpy(), 0), Tensor.from_numpy(sin.numpy(), 0), causal, tc)
    ref = r.taps["dec_layer0"].numpy()
    assert float(np.linalg.norm(y.numpy()[0] - ref) / np.linalg.norm(ref)) <= 1e-5
    # audio injection as inference.rs:114-124 does it (slice_scatter row by row) + embedding + last-row argmax
    emb = wts("thinker.model.embed_tokens.weight")
    ids, audio_pos = O.build_prompt(r.num_audio_tokens, None)
    hidden = Tensor.embedding(emb, Tensor.from_slice_i64(ids).to_device(0)).unsqueeze(0)
    ae = Tensor.from_numpy(r.taps["audio_embeds"].numpy(), 0)
    for i in range(3):
        hidden = hidden.slice_scatter(ae.narrow(0, i, 1).unsqueeze(0), 1, audio_pos[0] + i, audio_pos[0] + i + 1, 1)
    ref = r.taps["dec_embed"].numpy()
    got = hidden.numpy()[0]
    assert np.array_equal(got[:audio_pos[0] + 3], ref[:audio_pos[0] + 3])
    logits = Tensor.from_numpy(r.step_logits[0].numpy()[None, None], 0)
    assert logits.argmax(-1, False).int64_value([0, 0]) == r.all_step_ids[0]
