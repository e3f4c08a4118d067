//! HIP backend for MI355X (gfx950): bindings to libq3asr_hip.so.
//!
//! Two levels, both behind `#[cfg(feature = "hip")]`:
//!   * `engine`  -- the fused hot path (include/q3asr.h): what `AsrInference::transcribe` should call;
//!   * `ffi` / `array` / `ops` -- the op-level veneer (include/q3asr_ops.h) behind `struct Tensor`, so that the
//!     unchanged model code (mel.rs, audio_encoder.rs, layers.rs, text_decoder.rs) also runs, op by op.
//!
//! Not compiled in the build repository (no Rust toolchain on its boxes); `ffi.rs` is generated from the C header.
pub mod ffi;
pub mod array;
pub mod ops;
pub mod engine;
