"""MI355X-native Qwen3-ASR hot path (mel -> audio encoder -> Qwen3 decoder, greedy) behind the
C ABI declared in include/q3asr.h.  Python here is host glue only: ctypes bindings that mirror
the reference's L3/L4 interface (src/mel.rs, src/audio_encoder.rs, src/text_decoder.rs,
src/inference.rs).  All arithmetic runs in the hand-written HIP kernels of csrc/."""
__all__ = ["synthetic"]
