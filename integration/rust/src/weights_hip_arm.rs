//! The `hip` arm of the checkpoint loader (reference: src/weights.rs:61-131).  A maintainer pastes this function into
//! src/weights.rs next to the `tch-backend` (:62-120) and `mlx` (:123-131) arms; `load_model_weights` /
//! `load_sharded_safetensors` (:10-58) call `load_safetensors` by name and stay unchanged.
//!
//! Same contract as the tch arm: every tensor of the file ends up in the map as Float32 on `device` (the tch arm widens
//! BF16 / F16 on the host, weights.rs:74-89,134-181; here the raw bytes are copied to the device and widened there by
//! `q3a_op_to_dtype`), I64 stays I64, any other storage type is an error.  Uncompiled here (no Rust toolchain).

#[cfg(feature = "hip")]
pub fn load_safetensors(path: &Path, device: Device) -> Result<HashMap<String, Tensor>> {
    use crate::backend::hip::{array::HipArray, ffi};
    let dev_code: i32 = match device {
        Device::Cpu => ffi::Q3A_CPU,
        Device::Gpu(i) => i as i32,
    };
    let data =
        std::fs::read(path).with_context(|| format!("Failed to read safetensors: {:?}", path))?;
    let tensors = safetensors::SafeTensors::deserialize(&data)
        .with_context(|| format!("Failed to deserialize safetensors: {:?}", path))?;

    let mut result = HashMap::new();
    for (name, view) in tensors.iter() {
        let shape: Vec<i64> = view.shape().iter().map(|&s| s as i64).collect();
        let (code, widen) = match view.dtype() {
            safetensors::Dtype::BF16 => (ffi::Q3A_BF16, true),
            safetensors::Dtype::F16 => (ffi::Q3A_F16, true),
            safetensors::Dtype::F32 => (ffi::Q3A_F32, false),
            safetensors::Dtype::I64 => (ffi::Q3A_I64, false),
            dt => anyhow::bail!("Unsupported dtype in safetensors: {:?}", dt),
        };
        // one H2D copy of the stored bytes; BF16 / F16 are widened on the device (exact: both embed in f32)
        let raw = Tensor::from_hip(HipArray::from_bytes(view.data(), code, &shape, dev_code));
        let tensor = if widen { raw.to_dtype(DType::Float32) } else { raw };
        result.insert(name.to_string(), tensor);
    }
    Ok(result)
}
