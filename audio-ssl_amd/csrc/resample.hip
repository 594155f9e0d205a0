// f2 audio ingest: band-limited sinc resampling to the training rate, gfx950.
//
// Replaces the resampling half of `librosa.core.load(path, sr=16000)` (`src/dataset/upstream_dataset.py:55` of the reference;
// librosa 0.8.1 -> resampy 0.2.2 `resample(..., filter='kaiser_best')`): every output sample is a windowed-sinc interpolation
// of the input around t / ratio - left wing x[n - i], right wing x[n + 1 + k], taps read from a 64-zero-crossing Kaiser-windowed
// sinc table with 512 entries per zero crossing and linear interpolation between entries (interp_win + eta * interp_delta),
// table stride int(min(1, ratio) * 512).  resampy walks the output sequentially and adds every product into a float32 output
// element; here one thread owns one output sample and keeps that order and rounding (fp64 product, fp32 running sum), so the
// result is the sequential one.  The positions (n, table offsets, eta of both wings) come from the host, which accumulates the
// float64 time register exactly as resampy does (t / ratio by repeated addition) - one table per (length, rate pair), cached.
#include "common.h"

namespace {

struct ResampleArgs {
    const float* x; float* y;
    const int* n; const int* off_l; const int* off_r;       // [n_out]
    const double* eta_l; const double* eta_r;               // [n_out]
    const double* win; const double* delta;                 // [nwin]
    int n_orig, n_out, nwin, index_step;
};

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const long clip = blockIdx.y;
    if (t >= a.n_out) return;
    const float* x = a.x + clip * a.n_orig;
    const int n = a.n[t];
    float acc = 0.f;
    {
        const int off = a.off_l[t];
        const double eta = a.eta_l[t];
        const int i_max = min(n + 1, (a.nwin - off) / a.index_step);
        for (int i = 0; i < i_max; ++i) {
            const int j = off + i * a.index_step;
            const double w = a.win[j] + eta * a.delta[j];
            acc = (float)((double)acc + w * (double)x[n - i]);
        }
    }
    {
        const int off = a.off_r[t];
        const double eta = a.eta_r[t];
        const int k_max = min(a.n_orig - n - 1, (a.nwin - off) / a.index_step);
        for (int k = 0; k < k_max; ++k) {
            const int j = off + k * a.index_step;
            const double w = a.win[j] + eta * a.delta[j];
            acc = (float)((double)acc + w * (double)x[n + k + 1]);
        }
    }
    a.y[clip * a.n_out + t] = acc;
}

}  // namespace

// x [clips][n_orig] fp32 -> y [clips][n_out]; n / off_l / off_r int32 [n_out], eta_l / eta_r float64 [n_out], win / delta float64 [nwin].
extern "C" int audiossl_resample_sinc(const float* x, float* y, int clips, int n_orig, int n_out, const int* n, const int* off_l,
                                      const int* off_r, const double* eta_l, const double* eta_r, const double* win,
                                      const double* delta, int nwin, int index_step, void* stream) {
    ASSL_REQUIRE(x && y && n && off_l && off_r && eta_l && eta_r && win && delta);
    ASSL_REQUIRE(clips > 0 && n_orig > 0 && n_out > 0 && nwin > 0 && index_step > 0);
    ResampleArgs a{x, y, n, off_l, off_r, eta_l, eta_r, win, delta, n_orig, n_out, nwin, index_step};
    hipLaunchKernelGGL(resample_kernel, dim3(ceil_div(n_out, 256), clips), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    ASSL_LAUNCH_CHECK();
}
