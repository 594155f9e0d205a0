"""audio-ssl hot path on MI355X: reference-compatible `src` package (HIP kernels behind a C ABI)."""
