//! The third arm of `struct Tensor` (reference: src/tensor.rs).  A maintainer pastes these blocks into src/tensor.rs next
//! to the `tch-backend` (:145-488) and `mlx` (:492-954) arms; the struct gains
//!
//!     #[cfg(feature = "hip")]
//!     pub(crate) inner: crate::backend::hip::array::HipArray,
//!
//! and `Clone` gains `#[cfg(feature = "hip")] { Tensor { inner: self.inner.clone() } }`.
//! Every method below forwards to ONE `q3a_op_*` entry point (include/q3asr_ops.h) exactly as the tch arm forwards to ONE
//! libtorch call -- same names, same argument meaning, same panic-on-error behaviour.  Uncompiled here (no toolchain).

#[cfg(feature = "hip")]
use crate::backend::hip::{array::HipArray, ffi, ops};

#[cfg(feature = "hip")]
impl From<DType> for i32 {
    fn from(dt: DType) -> i32 {
        match dt {
            DType::Float32 => ffi::Q3A_F32,
            DType::Float16 => ffi::Q3A_F16,
            DType::BFloat16 => ffi::Q3A_BF16,
            DType::Int64 => ffi::Q3A_I64,
            DType::Int32 => ffi::Q3A_I32,
            DType::Bool => ffi::Q3A_BOOL,
        }
    }
}

#[cfg(feature = "hip")]
fn dtype_from_code(c: i32) -> DType {
    match c {
        ffi::Q3A_F16 => DType::Float16,
        ffi::Q3A_BF16 => DType::BFloat16,
        ffi::Q3A_I64 => DType::Int64,
        ffi::Q3A_I32 => DType::Int32,
        ffi::Q3A_BOOL => DType::Bool,
        _ => DType::Float32,
    }
}

#[cfg(feature = "hip")]
fn dev_code(d: Device) -> i32 {
    match d {
        Device::Cpu => ffi::Q3A_CPU,
        Device::Gpu(i) => i as i32,
    }
}

#[cfg(feature = "hip")]
#[allow(dead_code)]
impl Tensor {
    pub fn from_hip(a: HipArray) -> Self { Tensor { inner: a } }
    pub fn as_hip(&self) -> &HipArray { &self.inner }

    // -- Creation --
    pub fn from_slice_f32(data: &[f32]) -> Self { Tensor::from_hip(ops::from_slice_f32(data)) }
    pub fn from_slice_i64(data: &[i64]) -> Self { Tensor::from_hip(ops::from_slice_i64(data)) }
    pub fn zeros(shape: &[i64], dtype: DType, device: Device) -> Self { Tensor::from_hip(ops::zeros(shape, dtype.into(), dev_code(device))) }
    pub fn ones(shape: &[i64], dtype: DType, device: Device) -> Self { Tensor::from_hip(ops::ones(shape, dtype.into(), dev_code(device))) }
    pub fn full(shape: &[i64], val: f64, dtype: DType, device: Device) -> Self { Tensor::from_hip(ops::full(shape, val, dtype.into(), dev_code(device))) }
    pub fn arange(start: i64, end: i64, device: Device) -> Self { Tensor::from_hip(ops::arange(start, end, dev_code(device))) }
    pub fn arange_f(start: f64, end: f64, step: f64, dtype: DType, device: Device) -> Self {
        Tensor::from_hip(ops::arange_f(start, end, step, dtype.into(), dev_code(device)))
    }
    pub fn cat(tensors: &[Tensor], dim: i64) -> Self {
        let inner: Vec<&HipArray> = tensors.iter().map(|t| &t.inner).collect();
        Tensor::from_hip(ops::cat(&inner, dim))
    }
    pub fn stack(tensors: &[Tensor], dim: i64) -> Self {
        let inner: Vec<&HipArray> = tensors.iter().map(|t| &t.inner).collect();
        Tensor::from_hip(ops::stack(&inner, dim))
    }
    pub fn embedding(weight: &Tensor, indices: &Tensor) -> Self { Tensor::from_hip(ops::embedding(&weight.inner, &indices.inner)) }
    pub fn hann_window(size: i64, device: Device) -> Self { Tensor::from_hip(ops::hann_window(size, dev_code(device))) }

    // -- Shape --
    pub fn size(&self) -> Vec<i64> { self.inner.shape() }
    pub fn size3(&self) -> (i64, i64, i64) { let s = self.inner.shape(); assert_eq!(s.len(), 3); (s[0], s[1], s[2]) }
    pub fn size4(&self) -> (i64, i64, i64, i64) { let s = self.inner.shape(); assert_eq!(s.len(), 4); (s[0], s[1], s[2], s[3]) }
    pub fn dim(&self) -> usize { self.inner.ndim() }
    pub fn view(&self, shape: &[i64]) -> Self { Tensor::from_hip(ops::view(&self.inner, shape)) }
    pub fn reshape(&self, shape: &[i64]) -> Self { Tensor::from_hip(ops::reshape(&self.inner, shape)) }
    pub fn narrow(&self, dim: i64, start: i64, len: i64) -> Self { Tensor::from_hip(ops::narrow(&self.inner, dim, start, len)) }
    pub fn unsqueeze(&self, dim: i64) -> Self { Tensor::from_hip(ops::unsqueeze(&self.inner, dim)) }
    pub fn squeeze_dim(&self, dim: i64) -> Self { Tensor::from_hip(ops::squeeze_dim(&self.inner, dim)) }
    pub fn transpose(&self, dim0: i64, dim1: i64) -> Self { Tensor::from_hip(ops::transpose(&self.inner, dim0, dim1)) }
    pub fn permute(&self, dims: &[i64]) -> Self { Tensor::from_hip(ops::permute(&self.inner, dims)) }
    pub fn expand(&self, size: &[i64], _implicit: bool) -> Self { Tensor::from_hip(ops::expand(&self.inner, size)) }
    pub fn contiguous(&self) -> Self { Tensor::from_hip(ops::contiguous(&self.inner)) }
    pub fn tr(&self) -> Self { Tensor::from_hip(ops::tr(&self.inner)) }
    pub fn get(&self, index: i64) -> Self { Tensor::from_hip(ops::get(&self.inner, index)) }
    pub fn select(&self, dim: i64, index: i64) -> Self { Tensor::from_hip(ops::select(&self.inner, dim, index)) }

    // -- Arithmetic / math --
    pub fn matmul(&self, other: &Tensor) -> Self { Tensor::from_hip(ops::matmul(&self.inner, &other.inner)) }
    pub fn pow_scalar(&self, exp: f64) -> Self { Tensor::from_hip(ops::pow_scalar(&self.inner, exp)) }
    pub fn neg(&self) -> Self { Tensor::from_hip(ops::neg(&self.inner)) }
    pub fn clamp_min(&self, min: f64) -> Self { Tensor::from_hip(ops::clamp_min(&self.inner, min)) }
    pub fn maximum(&self, other: &Tensor) -> Self { Tensor::from_hip(ops::maximum(&self.inner, &other.inner)) }
    pub fn abs(&self) -> Self { Tensor::from_hip(ops::abs(&self.inner)) }
    pub fn square(&self) -> Self { Tensor::from_hip(ops::square(&self.inner)) }
    pub fn sqrt(&self) -> Self { Tensor::from_hip(ops::sqrt(&self.inner)) }
    pub fn rsqrt(&self) -> Self { Tensor::from_hip(ops::rsqrt(&self.inner)) }
    pub fn log10(&self) -> Self { Tensor::from_hip(ops::log10(&self.inner)) }
    pub fn sin(&self) -> Self { Tensor::from_hip(ops::sin(&self.inner)) }
    pub fn cos(&self) -> Self { Tensor::from_hip(ops::cos(&self.inner)) }
    pub fn exp(&self) -> Self { Tensor::from_hip(ops::exp(&self.inner)) }
    pub fn softmax(&self, dim: i64) -> Self { Tensor::from_hip(ops::softmax(&self.inner, dim)) }
    pub fn gelu(&self) -> Self { Tensor::from_hip(ops::gelu(&self.inner)) }
    pub fn silu(&self) -> Self { Tensor::from_hip(ops::silu(&self.inner)) }
    pub fn mean_dim(&self, dims: &[i64], keepdim: bool) -> Self { Tensor::from_hip(ops::mean_dim(&self.inner, dims, keepdim as i32)) }
    pub fn max(&self) -> Self { Tensor::from_hip(ops::max(&self.inner)) }
    pub fn argmax(&self, dim: i64, keepdim: bool) -> Self { Tensor::from_hip(ops::argmax(&self.inner, dim, keepdim as i32)) }
    pub fn triu(&self, diagonal: i64) -> Self { Tensor::from_hip(ops::triu(&self.inner, diagonal)) }
    pub fn slice_scatter(&self, src: &Tensor, dim: i64, start: i64, end: i64, step: i64) -> Self {
        Tensor::from_hip(ops::slice_scatter(&self.inner, &src.inner, dim, start, end, step))
    }
    pub fn fill_(&mut self, val: f64) { ops::fill_inplace(&mut self.inner, val) }
    pub fn layer_norm(&self, normalized_shape: &[i64], weight: Option<&Tensor>, bias: Option<&Tensor>, eps: f64) -> Self {
        Tensor::from_hip(ops::layer_norm(&self.inner, normalized_shape, weight.map(|w| &w.inner), bias.map(|b| &b.inner), eps))
    }
    pub fn conv2d(&self, weight: &Tensor, bias: Option<&Tensor>, stride: &[i64], padding: &[i64], dilation: &[i64], groups: i64) -> Self {
        Tensor::from_hip(ops::conv2d(&self.inner, &weight.inner, bias.map(|b| &b.inner), stride, padding, dilation, groups))
    }
    pub fn reflection_pad1d(&self, pad: &[i64]) -> Self { Tensor::from_hip(ops::reflection_pad1d(&self.inner, pad)) }
    pub fn stft(&self, n_fft: i64, hop_length: i64, win_length: i64, window: &Tensor, normalized: bool, onesided: bool,
                return_complex: bool) -> Self {
        Tensor::from_hip(ops::stft(&self.inner, n_fft, hop_length, win_length, &window.inner, normalized as i32, onesided as i32,
                                   return_complex as i32))
    }

    // -- Type / device / extraction --
    pub fn to_dtype(&self, dtype: DType) -> Self { Tensor::from_hip(ops::to_dtype(&self.inner, dtype.into())) }
    pub fn to_device(&self, device: Device) -> Self { Tensor::from_hip(ops::to_device(&self.inner, dev_code(device))) }
    pub fn kind(&self) -> DType { dtype_from_code(self.inner.dtype()) }
    pub fn device(&self) -> Device { match self.inner.device() { d if d < 0 => Device::Cpu, d => Device::Gpu(d as usize) } }
    pub fn shallow_clone(&self) -> Self { Tensor::from_hip(self.inner.clone()) }
    pub fn int64_value(&self, indices: &[i64]) -> i64 { self.inner.int64_value(indices) }
    pub fn f64_value(&self, indices: &[i64]) -> f64 { self.inner.f64_value(indices) }
    pub fn to_vec_f32(&self) -> Vec<f32> { self.inner.to_vec_f32() }
}

// Operator arms (reference: src/tensor.rs:960-1161).  Each existing `impl std::ops::X<..> for &Tensor` body gains
//
//     #[cfg(feature = "hip")]
//     { Tensor::from_hip(crate::backend::hip::ops::add(&self.inner, &rhs.inner)) }          // Tensor + Tensor   (sub / mul / div alike)
//     #[cfg(feature = "hip")]
//     { Tensor::from_hip(crate::backend::hip::ops::add_scalar(&self.inner, rhs)) }           // Tensor + f64      (sub_scalar / mul_scalar / div_scalar)
//     #[cfg(feature = "hip")]
//     { Tensor::from_hip(crate::backend::hip::ops::neg(&self.inner)) }                       // unary -
//     #[cfg(feature = "hip")]
//     { crate::backend::hip::ops::add_inplace(&mut self.inner, &rhs.inner); }                // +=
//
// The other places the backend leaks have their own files next to this one:
//   * src/weights.rs:61-131        -> weights_hip_arm.rs        (`load_safetensors` arm)
//   * src/audio_encoder.rs:227-259 -> audio_encoder_hip_arm.rs  (window mask without `where_self`)
//   * src/main.rs:51-65, lib.rs, Cargo.toml, build.rs -> main_hip_arm.rs
