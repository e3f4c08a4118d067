//! The `hip` arm of the attention-window mask (reference: src/audio_encoder.rs:172-260, the backend-specific tail at
//! :227-259).  A maintainer pastes the block below into `AudioEncoder::build_window_mask` after the `mlx` block; the
//! host loop that fills `allow_data` (:188-210) stays as it is.
//!
//! The tch arm needs `where_self`, which is not part of the `Tensor` surface; this arm does not: the additive mask is
//! written on the host as 0.0 / -inf and uploaded with the two `Tensor` calls every arm already has.  (With the
//! engine-level binding, src/backend/hip/engine.rs, no mask tensor exists at all: the windows are segments of the fused
//! attention kernel.)  Uncompiled here (no Rust toolchain).

        #[cfg(feature = "hip")]
        {
            let _ = (&neg_inf, &zero); // the two full-size temporaries of the other arms are not needed here
            let mask_vals: Vec<f32> = allow_data
                .iter()
                .map(|&allowed| if allowed { 0.0f32 } else { f32::NEG_INFINITY })
                .collect();
            let mask = Tensor::from_slice_f32(&mask_vals)
                .reshape(&[1, 1, total_tokens, total_tokens])
                .to_device(device);
            Some(mask)
        }
