"""Python mirror of the reference's backend seam `struct Tensor` (src/tensor.rs:120-126, tch arm :145-488, operators
:960-1161) over the op-level C ABI of include/q3asr_ops.h -- the same method names and argument meaning, so that the
op-level parity tests read like calls into the reference's own Tensor.  A Rust `#[cfg(feature = "hip")]` arm binds the very
same entry points (integration/rust/src/backend/hip/).  No arithmetic here: every number comes from libq3asr_hip.so.
"""
from __future__ import annotations

import ctypes as C
import re
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

F32, F16, BF16, I64, I32, BOOL, C64 = range(7)
CPU = -1
_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "q3asr_ops.h")

_ARG = {"q3a_array**": C.POINTER(C.c_void_p), "const q3a_array*": C.c_void_p, "q3a_array*": C.c_void_p, "q3a_stream*": C.c_void_p,
        "const q3a_array* const*": C.POINTER(C.c_void_p), "int64_t": C.c_int64, "int32_t": C.c_int32, "double": C.c_double,
        "const int64_t*": C.POINTER(C.c_int64), "int64_t*": C.POINTER(C.c_int64), "const float*": C.POINTER(C.c_float),
        "float*": C.POINTER(C.c_float), "double*": C.POINTER(C.c_double), "const void*": C.c_void_p, "void": None,
        "const char*": C.c_char_p}
_lib_ops = None


def declared_symbols() -> List[str]:
    text = re.sub(r"/\*.*?\*/", "", open(_HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(q3a_[a-z0-9_]+)\s*\(", text)))


def lib():
    """libq3asr_hip.so with the argtypes of every function include/q3asr_ops.h declares (parsed from the header)."""
    global _lib_ops
    if _lib_ops is not None:
        return _lib_ops
    L = _lib.load()
    text = re.sub(r"/\*.*?\*/", "", open(_HDR).read(), flags=re.S)
    for m in re.finditer(r"^\s*(const char\*|int32_t|int64_t|void)\s+(q3a_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.M):
        rt, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        fn = getattr(L, name)
        fn.restype = {"const char*": C.c_char_p, "int32_t": C.c_int32, "int64_t": C.c_int64, "void": None}[rt]
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                ty = re.sub(r"\s*\b[a-z_0-9]+$", "", a).strip() if not a.endswith("*") else a
                at.append(_ARG[ty])
        fn.argtypes = at
    _lib_ops = L
    return L


class OpsError(RuntimeError):
    pass


def _i64(v: Sequence[int]):
    a = (C.c_int64 * max(len(v), 1))(*[int(x) for x in v])
    return a


class Tensor:
    """One q3a_array handle (RAII: freed in __del__, as Drop frees it on the Rust side)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self._h and self._h.value:
                lib().q3a_array_free(self._h)
        except Exception:
            pass

    # ---- plumbing ----
    @staticmethod
    def _call(name: str, *args) -> "Tensor":
        out = C.c_void_p()
        if getattr(lib(), name)(C.byref(out), *args) != 0:
            raise OpsError((lib().q3a_ops_last_error() or b"").decode())
        return Tensor(out.value)

    @staticmethod
    def _chk(rc: int):
        if rc != 0:
            raise OpsError((lib().q3a_ops_last_error() or b"").decode())

    # ---- creation (tensor.rs:162-219) ----
    @staticmethod
    def from_slice_f32(data) -> "Tensor":
        a = np.ascontiguousarray(data, dtype=np.float32)
        return Tensor._call("q3a_op_from_slice_f32", a.ctypes.data_as(C.POINTER(C.c_float)), a.size)

    @staticmethod
    def from_slice_i64(data) -> "Tensor":
        a = np.ascontiguousarray(data, dtype=np.int64)
        return Tensor._call("q3a_op_from_slice_i64", a.ctypes.data_as(C.POINTER(C.c_int64)), a.size)

    @staticmethod
    def from_numpy(a: np.ndarray, device: int = 0) -> "Tensor":
        """weights.rs loader arm: raw bytes + dtype + shape straight to the device."""
        dt = {np.dtype(np.float32): F32, np.dtype(np.int64): I64, np.dtype(np.int32): I32, np.dtype(np.bool_): BOOL}[a.dtype]
        a = np.ascontiguousarray(a)
        return Tensor._call("q3a_op_from_bytes", a.ctypes.data_as(C.c_void_p), dt, _i64(a.shape), a.ndim, device)

    @staticmethod
    def zeros(shape, dtype=F32, device=0): return Tensor._call("q3a_op_zeros", _i64(shape), len(shape), dtype, device)
    @staticmethod
    def ones(shape, dtype=F32, device=0): return Tensor._call("q3a_op_ones", _i64(shape), len(shape), dtype, device)
    @staticmethod
    def full(shape, val, dtype=F32, device=0): return Tensor._call("q3a_op_full", _i64(shape), len(shape), float(val), dtype, device)
    @staticmethod
    def arange(start, end, device=0): return Tensor._call("q3a_op_arange", start, end, device)
    @staticmethod
    def arange_f(start, end, step, dtype=F32, device=0): return Tensor._call("q3a_op_arange_f", float(start), float(end), float(step), dtype, device)
    @staticmethod
    def hann_window(size, device=0): return Tensor._call("q3a_op_hann_window", size, device)

    @staticmethod
    def cat(tensors: Sequence["Tensor"], dim: int) -> "Tensor":
        arr = (C.c_void_p * len(tensors))(*[t._h.value for t in tensors])
        return Tensor._call("q3a_op_cat", arr, len(tensors), dim, None)

    @staticmethod
    def stack(tensors: Sequence["Tensor"], dim: int) -> "Tensor":
        arr = (C.c_void_p * len(tensors))(*[t._h.value for t in tensors])
        return Tensor._call("q3a_op_stack", arr, len(tensors), dim, None)

    @staticmethod
    def embedding(weight: "Tensor", indices: "Tensor") -> "Tensor":
        return Tensor._call("q3a_op_embedding", weight._h, indices._h, None)

    # ---- shape (tensor.rs:223-285) ----
    def size(self) -> List[int]:
        buf = (C.c_int64 * 8)()
        n = lib().q3a_array_shape(self._h, buf, 8)
        return [int(buf[i]) for i in range(n)]

    def dim(self) -> int: return int(lib().q3a_array_ndim(self._h))
    def kind(self) -> int: return int(lib().q3a_array_dtype(self._h))
    def device(self) -> int: return int(lib().q3a_array_device(self._h))
    def view(self, shape): return Tensor._call("q3a_op_view", self._h, _i64(shape), len(shape))
    def reshape(self, shape): return Tensor._call("q3a_op_reshape", self._h, _i64(shape), len(shape), None)
    def narrow(self, dim, start, length): return Tensor._call("q3a_op_narrow", self._h, dim, start, length)
    def unsqueeze(self, dim): return Tensor._call("q3a_op_unsqueeze", self._h, dim)
    def squeeze_dim(self, dim): return Tensor._call("q3a_op_squeeze_dim", self._h, dim)
    def transpose(self, d0, d1): return Tensor._call("q3a_op_transpose", self._h, d0, d1)
    def permute(self, dims): return Tensor._call("q3a_op_permute", self._h, _i64(dims), len(dims))
    def expand(self, size, implicit=False): return Tensor._call("q3a_op_expand", self._h, _i64(size), len(size))
    def contiguous(self): return Tensor._call("q3a_op_contiguous", self._h, None)
    def tr(self): return Tensor._call("q3a_op_tr", self._h)
    def get(self, index): return Tensor._call("q3a_op_get", self._h, index)
    def select(self, dim, index): return Tensor._call("q3a_op_select", self._h, dim, index)
    def shallow_clone(self): return Tensor._call("q3a_op_shallow_clone", self._h)

    # ---- arithmetic / math (tensor.rs:289-372, operators :960-1161) ----
    def matmul(self, other): return Tensor._call("q3a_op_matmul", self._h, other._h, None)
    def pow_scalar(self, e): return Tensor._call("q3a_op_pow_scalar", self._h, float(e), None)
    def neg(self): return Tensor._call("q3a_op_neg", self._h, None)
    def clamp_min(self, m): return Tensor._call("q3a_op_clamp_min", self._h, float(m), None)
    def maximum(self, other): return Tensor._call("q3a_op_maximum", self._h, other._h, None)
    def abs(self): return Tensor._call("q3a_op_abs", self._h, None)
    def square(self): return Tensor._call("q3a_op_square", self._h, None)
    def sqrt(self): return Tensor._call("q3a_op_sqrt", self._h, None)
    def rsqrt(self): return Tensor._call("q3a_op_rsqrt", self._h, None)
    def log10(self): return Tensor._call("q3a_op_log10", self._h, None)
    def sin(self): return Tensor._call("q3a_op_sin", self._h, None)
    def cos(self): return Tensor._call("q3a_op_cos", self._h, None)
    def exp(self): return Tensor._call("q3a_op_exp", self._h, None)
    def softmax(self, dim): return Tensor._call("q3a_op_softmax", self._h, dim, None)
    def gelu(self): return Tensor._call("q3a_op_gelu", self._h, None)
    def silu(self): return Tensor._call("q3a_op_silu", self._h, None)
    def mean_dim(self, dims, keepdim): return Tensor._call("q3a_op_mean_dim", self._h, _i64(dims), len(dims), int(keepdim), None)
    def max(self): return Tensor._call("q3a_op_max", self._h, None)
    def argmax(self, dim, keepdim): return Tensor._call("q3a_op_argmax", self._h, dim, int(keepdim), None)
    def triu(self, diagonal): return Tensor._call("q3a_op_triu", self._h, diagonal, None)
    def slice_scatter(self, src, dim, start, end, step): return Tensor._call("q3a_op_slice_scatter", self._h, src._h, dim, start, end, step, None)

    def fill_(self, val):
        Tensor._chk(lib().q3a_op_fill_inplace(self._h, float(val), None))

    def layer_norm(self, normalized_shape, weight: Optional["Tensor"], bias: Optional["Tensor"], eps: float):
        return Tensor._call("q3a_op_layer_norm", self._h, _i64(normalized_shape), len(normalized_shape),
                            weight._h if weight is not None else None, bias._h if bias is not None else None, float(eps), None)

    def conv2d(self, weight, bias, stride, padding, dilation, groups):
        return Tensor._call("q3a_op_conv2d", self._h, weight._h, bias._h if bias is not None else None, _i64(stride), _i64(padding),
                            _i64(dilation), groups, None)

    def reflection_pad1d(self, pad): return Tensor._call("q3a_op_reflection_pad1d", self._h, _i64(pad), None)

    def stft(self, n_fft, hop_length, win_length, window, normalized, onesided, return_complex):
        return Tensor._call("q3a_op_stft", self._h, n_fft, hop_length, win_length, window._h, int(normalized), int(onesided),
                            int(return_complex), None)

    def to_dtype(self, dtype): return Tensor._call("q3a_op_to_dtype", self._h, dtype, None)
    def to_device(self, device): return Tensor._call("q3a_op_to_device", self._h, device, None)

    def int64_value(self, indices) -> int:
        v = C.c_int64()
        Tensor._chk(lib().q3a_array_int64_value(self._h, _i64(indices), len(indices), C.byref(v)))
        return int(v.value)

    def f64_value(self, indices) -> float:
        v = C.c_double()
        Tensor._chk(lib().q3a_array_f64_value(self._h, _i64(indices), len(indices), C.byref(v)))
        return float(v.value)

    def to_vec_f32(self) -> np.ndarray:
        n = int(lib().q3a_array_numel(self._h))
        out = np.zeros(max(n, 1), dtype=np.float32)
        Tensor._chk(lib().q3a_array_to_vec_f32(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out[:n]

    def numpy(self) -> np.ndarray:
        return self.to_vec_f32().reshape(self.size())

    # operators (tensor.rs:960-1161)
    def _bin(self, other, op, sop):
        if isinstance(other, Tensor):
            return Tensor._call(op, self._h, other._h, None)
        return Tensor._call(sop, self._h, float(other), None)

    def __add__(self, o): return self._bin(o, "q3a_op_add", "q3a_op_add_scalar")
    def __sub__(self, o): return self._bin(o, "q3a_op_sub", "q3a_op_sub_scalar")
    def __mul__(self, o): return self._bin(o, "q3a_op_mul", "q3a_op_mul_scalar")
    def __truediv__(self, o): return self._bin(o, "q3a_op_div", "q3a_op_div_scalar")
    def __neg__(self): return self.neg()

    def __iadd__(self, o):
        Tensor._chk(lib().q3a_op_add_inplace(self._h, o._h, None))
        return self
