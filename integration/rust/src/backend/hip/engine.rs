//! Engine-level binding (include/q3asr.h): the fused hot path.  `AsrInference::transcribe` under `feature = "hip"` calls
//! `HipEngine::transcribe_batch` (steps 2-8 of src/inference.rs:95-200 in one FFI call) instead of walking the model
//! through `Tensor` ops; `HipGroup` is the same for several GPUs (one broadcast of the weights, utterances partitioned).
//! Uncompiled here (no Rust toolchain on the build boxes).
#![allow(non_camel_case_types, dead_code)]

use anyhow::{bail, Result};
use std::ffi::{CStr, CString};
use std::os::raw::c_char;
use std::path::Path;

#[repr(C)]
pub struct q3a_engine { _private: [u8; 0] }
#[repr(C)]
pub struct q3a_group { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct q3a_opts {
    pub precise: i32,
    pub max_new_tokens: i32,
    pub use_graph: i32,
    pub debug_taps: i32,
    pub valu_attention: i32,
    pub reserved: [i32; 11],
}

#[link(name = "q3asr_hip")]
extern "C" {
    pub fn q3a_opts_default(o: *mut q3a_opts);
    pub fn q3a_device_count() -> i32;
    pub fn q3a_engine_create(model_dir: *const c_char, device: i32, opts: *const q3a_opts, out: *mut *mut q3a_engine) -> i32;
    pub fn q3a_engine_destroy(e: *mut q3a_engine);
    pub fn q3a_last_error(e: *const q3a_engine) -> *const c_char;
    pub fn q3a_transcribe_batch(e: *mut q3a_engine, pcm16k: *const f32, n_samples: *const i64, b: i32, lang_prefix_ids: *const i32,
                                n_prefix: i32, max_new: i32, fixed_new_tokens: i32, out_ids: *mut i32, stride: i32, out_lens: *mut i32) -> i32;
    pub fn q3a_transcribe_batch_ptrs(e: *mut q3a_engine, pcm16k: *const *const f32, n_samples: *const i64, b: i32, lang_prefix_ids: *const i32,
                                     n_prefix: i32, max_new: i32, fixed_new_tokens: i32, out_ids: *mut i32, stride: i32, out_lens: *mut i32) -> i32;
    pub fn q3a_group_create(model_dir: *const c_char, n_gpus: i32, devices: *const i32, opts: *const q3a_opts, out: *mut *mut q3a_group) -> i32;
    pub fn q3a_group_destroy(g: *mut q3a_group);
    pub fn q3a_group_size(g: *const q3a_group) -> i32;
    pub fn q3a_group_last_error(g: *const q3a_group) -> *const c_char;
    pub fn q3a_group_startup_seconds(g: *const q3a_group, out4: *mut f64) -> i32;
    pub fn q3a_group_transcribe(g: *mut q3a_group, pcm16k: *const f32, n_samples: *const i64, b: i32, lang_prefix_ids: *const i32,
                                n_prefix: i32, max_new: i32, fixed_new_tokens: i32, out_ids: *mut i32, stride: i32, out_lens: *mut i32) -> i32;
    pub fn q3a_group_transcribe_ptrs(g: *mut q3a_group, pcm16k: *const *const f32, n_samples: *const i64, b: i32, lang_prefix_ids: *const i32,
                                     n_prefix: i32, max_new: i32, fixed_new_tokens: i32, out_ids: *mut i32, stride: i32, out_lens: *mut i32) -> i32;
}

/// src/main.rs:51-65 for the `hip` feature: HIP devices visible to this process (0: none -- there is no CPU path).
pub fn device_count() -> i32 { unsafe { q3a_device_count() } }

fn msg(p: *const c_char) -> String {
    if p.is_null() { "unknown error".into() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}

/// One GPU: owns a `q3a_engine` (weights arena, workspace, stream).  One host thread per handle.
pub struct HipEngine { raw: *mut q3a_engine }
unsafe impl Send for HipEngine {}

impl Drop for HipEngine {
    fn drop(&mut self) { if !self.raw.is_null() { unsafe { q3a_engine_destroy(self.raw) } } }
}

impl HipEngine {
    /// AsrInference::load's model part (src/inference.rs:30-66): config.json + safetensors -> device arena.
    pub fn load(model_dir: &Path, device: usize) -> Result<Self> {
        let dir = CString::new(model_dir.to_string_lossy().as_bytes())?;
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { q3a_engine_create(dir.as_ptr(), device as i32, std::ptr::null(), &mut raw) };
        if rc != 0 { bail!("Failed to load model: {}", msg(unsafe { q3a_last_error(std::ptr::null()) })); }
        Ok(HipEngine { raw })
    }

    /// Steps 2-8 of `transcribe` for a batch of independent utterances (16 kHz f32).  Returns generated ids (EOS excluded).
    pub fn transcribe_batch(&self, clips: &[&[f32]], lang_prefix_ids: &[i32], max_new: usize) -> Result<Vec<Vec<i64>>> {
        let b = clips.len();
        let n: Vec<i64> = clips.iter().map(|c| c.len() as i64).collect();
        // one pointer per utterance: the Vec<f32> that load_audio (src/audio.rs:7) returned for each file is handed over as it is --
        // no concatenation on the host; the library stages the pageable buffers into pinned memory and overlaps the copy with the mel
        let ptrs: Vec<*const f32> = clips.iter().map(|c| c.as_ptr()).collect();
        let mut ids = vec![0i32; b * max_new];
        let mut lens = vec![0i32; b];
        let pre = if lang_prefix_ids.is_empty() { std::ptr::null() } else { lang_prefix_ids.as_ptr() };
        let rc = unsafe {
            q3a_transcribe_batch_ptrs(self.raw, ptrs.as_ptr(), n.as_ptr(), b as i32, pre, lang_prefix_ids.len() as i32, max_new as i32, 0,
                                      ids.as_mut_ptr(), max_new as i32, lens.as_mut_ptr())
        };
        if rc != 0 { bail!("{}", msg(unsafe { q3a_last_error(self.raw) })); }
        Ok((0..b).map(|i| ids[i * max_new..i * max_new + lens[i] as usize].iter().map(|&x| x as i64).collect()).collect())
    }
}

/// Several GPUs of one node: weights read once, one RCCL broadcast, utterances partitioned contiguously.
pub struct HipGroup { raw: *mut q3a_group }
unsafe impl Send for HipGroup {}

impl Drop for HipGroup {
    fn drop(&mut self) { if !self.raw.is_null() { unsafe { q3a_group_destroy(self.raw) } } }
}

impl HipGroup {
    pub fn load(model_dir: &Path, n_gpus: usize) -> Result<Self> {
        let dir = CString::new(model_dir.to_string_lossy().as_bytes())?;
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { q3a_group_create(dir.as_ptr(), n_gpus as i32, std::ptr::null(), std::ptr::null(), &mut raw) };
        if rc != 0 { bail!("Failed to load model: {}", msg(unsafe { q3a_last_error(std::ptr::null()) })); }
        Ok(HipGroup { raw })
    }

    /// Start-up stage times in seconds: [checkpoint read + pack into pinned memory, H2D upload, RCCL broadcast, engine creation].
    pub fn startup_seconds(&self) -> [f64; 4] {
        let mut v = [0f64; 4];
        unsafe { q3a_group_startup_seconds(self.raw, v.as_mut_ptr()) };
        v
    }

    pub fn transcribe_batch(&self, clips: &[&[f32]], lang_prefix_ids: &[i32], max_new: usize) -> Result<Vec<Vec<i64>>> {
        let b = clips.len();
        let n: Vec<i64> = clips.iter().map(|c| c.len() as i64).collect();
        let ptrs: Vec<*const f32> = clips.iter().map(|c| c.as_ptr()).collect();
        let mut ids = vec![0i32; b * max_new];
        let mut lens = vec![0i32; b];
        let pre = if lang_prefix_ids.is_empty() { std::ptr::null() } else { lang_prefix_ids.as_ptr() };
        let rc = unsafe {
            q3a_group_transcribe_ptrs(self.raw, ptrs.as_ptr(), n.as_ptr(), b as i32, pre, lang_prefix_ids.len() as i32, max_new as i32, 0,
                                      ids.as_mut_ptr(), max_new as i32, lens.as_mut_ptr())
        };
        if rc != 0 { bail!("{}", msg(unsafe { q3a_group_last_error(self.raw) })); }
        Ok((0..b).map(|i| ids[i * max_new..i * max_new + lens[i] as usize].iter().map(|&x| x as i64).collect()).collect())
    }
}
