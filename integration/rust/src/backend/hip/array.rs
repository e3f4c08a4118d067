//! Safe RAII wrapper around a `q3a_array` handle (the role `MlxArray` plays for the MLX arm, src/backend/mlx/array.rs).

use super::ffi;
use std::ffi::CStr;
use std::fmt;

/// Owned handle to a reference-counted device (or host) array.  `Clone` is a shallow clone (shared storage),
/// `Drop` releases the handle -- the semantics `struct Tensor` expects of its `inner` (src/tensor.rs:134-141).
pub struct HipArray {
    pub(crate) ptr: *mut ffi::q3a_array,
}

// The library serialises a device's work on one in-order stream; handles may move between threads.
unsafe impl Send for HipArray {}
unsafe impl Sync for HipArray {}

impl Drop for HipArray {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { ffi::q3a_array_free(self.ptr) };
        }
    }
}

impl Clone for HipArray {
    fn clone(&self) -> Self {
        let mut out = std::ptr::null_mut();
        check(unsafe { ffi::q3a_op_shallow_clone(&mut out, self.ptr) });
        HipArray { ptr: out }
    }
}

impl fmt::Debug for HipArray {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "HipArray(shape={:?}, dtype={})", self.shape(), self.dtype())
    }
}

/// The tch arm's forward paths are infallible and panic on shape / dtype errors (src/tensor.rs:228,232): same here.
#[inline]
pub(crate) fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(ffi::q3a_ops_last_error()) }.to_string_lossy().into_owned();
        panic!("q3asr_hip: {}", msg);
    }
}

impl HipArray {
    pub(crate) fn from_raw(ptr: *mut ffi::q3a_array) -> Self {
        HipArray { ptr }
    }

    pub fn from_f32(data: &[f32]) -> Self {
        let mut out = std::ptr::null_mut();
        check(unsafe { ffi::q3a_op_from_slice_f32(&mut out, data.as_ptr(), data.len() as i64) });
        HipArray { ptr: out }
    }

    pub fn from_i64(data: &[i64]) -> Self {
        let mut out = std::ptr::null_mut();
        check(unsafe { ffi::q3a_op_from_slice_i64(&mut out, data.as_ptr(), data.len() as i64) });
        HipArray { ptr: out }
    }

    /// weights.rs loader arm: raw safetensors bytes of `dtype` and `shape`, copied to `device` (-1 = host).
    pub fn from_bytes(data: &[u8], dtype: i32, shape: &[i64], device: i32) -> Self {
        let mut out = std::ptr::null_mut();
        check(unsafe {
            ffi::q3a_op_from_bytes(&mut out, data.as_ptr() as *const _, dtype, shape.as_ptr(), shape.len() as i32, device)
        });
        HipArray { ptr: out }
    }

    pub fn shape(&self) -> Vec<i64> {
        let mut buf = [0i64; 8];
        let n = unsafe { ffi::q3a_array_shape(self.ptr, buf.as_mut_ptr(), 8) };
        buf[..n.max(0) as usize].to_vec()
    }

    pub fn ndim(&self) -> usize {
        unsafe { ffi::q3a_array_ndim(self.ptr) as usize }
    }

    pub fn dtype(&self) -> i32 {
        unsafe { ffi::q3a_array_dtype(self.ptr) }
    }

    pub fn device(&self) -> i32 {
        unsafe { ffi::q3a_array_device(self.ptr) }
    }

    pub fn numel(&self) -> usize {
        unsafe { ffi::q3a_array_numel(self.ptr) as usize }
    }

    pub fn int64_value(&self, indices: &[i64]) -> i64 {
        let mut v = 0i64;
        check(unsafe { ffi::q3a_array_int64_value(self.ptr, indices.as_ptr(), indices.len() as i32, &mut v) });
        v
    }

    pub fn f64_value(&self, indices: &[i64]) -> f64 {
        let mut v = 0f64;
        check(unsafe { ffi::q3a_array_f64_value(self.ptr, indices.as_ptr(), indices.len() as i32, &mut v) });
        v
    }

    pub fn to_vec_f32(&self) -> Vec<f32> {
        let n = self.numel();
        let mut out = vec![0f32; n];
        check(unsafe { ffi::q3a_array_to_vec_f32(self.ptr, out.as_mut_ptr(), n as i64) });
        out
    }
}
