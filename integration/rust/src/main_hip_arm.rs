//! The `hip` arm of the device selection in the CLI (reference: src/main.rs:51-65) and of the crate-level switches
//! (src/lib.rs:2-10, src/error.rs:23-25, Cargo.toml:30-35, build.rs:9-12).  Uncompiled here (no Rust toolchain).

// ---- src/main.rs, after the `mlx` block (:59-65) ----
    #[cfg(feature = "hip")]
    let device = {
        let n = qwen3_asr::backend::hip::engine::device_count();
        if n <= 0 {
            anyhow::bail!("No HIP device available (libq3asr_hip has no CPU path)");
        }
        tracing::info!("Using HIP device 0 of {} (MI355X / gfx950)", n);
        Device::Gpu(0)
    };

// ---- src/lib.rs, next to the exclusivity checks (:2-10) ----
#[cfg(any(all(feature = "hip", feature = "tch-backend"), all(feature = "hip", feature = "mlx")))]
compile_error!("Features `hip`, `tch-backend` and `mlx` are mutually exclusive.");
#[cfg(not(any(feature = "tch-backend", feature = "mlx", feature = "hip")))]
compile_error!("One of `tch-backend`, `mlx` or `hip` must be enabled.");

// ---- src/error.rs, next to the tch conversion (:23-25): the op-level arm panics like tch's non-`f_` methods; the
// engine-level calls return `anyhow::Error` built from `q3a_last_error` (src/backend/hip/engine.rs), so no new variant
// is needed ----

// ---- Cargo.toml [features] (:30-35) ----
// hip = []            # links libq3asr_hip.so; no Rust dependencies of its own

// ---- build.rs (:9-12), next to the mlx link directives ----
//     #[cfg(feature = "hip")]
//     {
//         let dir = std::env::var("Q3ASR_HIP_LIB_DIR").expect("set Q3ASR_HIP_LIB_DIR to the directory of libq3asr_hip.so");
//         println!("cargo:rustc-link-search=native={}", dir);
//         println!("cargo:rustc-link-lib=dylib=q3asr_hip");
//         println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
//     }
