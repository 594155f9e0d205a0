// K1: waveform -> log-mel spectrogram, gfx950.
//
// Replaces the per-clip CPU chain of the reference (`src/utils/utils.py:26-28, 43-49`:
// librosa.stft(n_fft=1024, hop) -> |X|^2 -> mel filterbank matmul -> log(. + eps)) with one launch over
// the whole batch.  One 64-lane wave owns one frame: the 1024 real samples (reflect-padded, periodic
// Hann) are packed as 512 complex points, 8 per lane, and transformed by a Stockham radix-8 FFT
// (3 passes; pass 0 straight from registers, two exchanges through padded LDS), then split into the
// 513-bin real spectrum, squared, reduced by the sparse mel rows (<=45 taps each for the default
// filterbank) held in LDS, and logged.  A 256-thread block handles 8 frames of one clip and writes an
// [n_mels][8] tile so the global stores are 32-byte runs.
// HBM roofline: 4*L bytes read + 4*n_mels*T bytes written per clip (89,856 B at L=16000, T=101).
#include "common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int FPB = 8;                 // frames per block
constexpr int CPAD = 512 + 64;         // padded complex-buffer length: idx + (idx >> 3)

__device__ __forceinline__ int pidx(int a) { return a + (a >> 3); }

struct C32 { float x, y; };
__device__ __forceinline__ C32 cadd(C32 a, C32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ C32 csub(C32 a, C32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ C32 cmul(C32 a, C32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ C32 mul_mi(C32 a) { return {a.y, -a.x}; }     // * (-i)

// 8-point forward DFT, natural order in / out.
__device__ __forceinline__ void dft8(C32* v) {
    const float s = 0.70710678118654752440f;
    C32 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    C32 a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    C32 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    C32 a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    C32 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    C32 b4 = cadd(a4, a6), b6 = mul_mi(csub(a4, a6));
    C32 t5 = cadd(a5, a7), t7 = csub(a5, a7);
    C32 b5 = {(t5.x + t5.y) * s, (t5.y - t5.x) * s};          // * e^{-i pi/4}
    C32 b7 = {(t7.y - t7.x) * s, (-t7.x - t7.y) * s};         // * e^{-3i pi/4}
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cadd(b1, b5); v[5] = csub(b1, b5);
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[3] = cadd(b3, b7); v[7] = csub(b3, b7);
}

struct LogmelArgs {
    const float* wave; float* out;
    int B, L, T, hop, n_mels, taps;
    const float* win;        // [1024] periodic Hann
    const float* tw;         // [1024][2] exp(-2 pi i k / 1024)
    const float* melw;       // [n_mels][taps] packed non-zero run of each filterbank row
    const int* mel_start;    // [n_mels] first bin of the run
    float eps_pow, eps_log;
    int apply_log;           // 0: return the mel power (MelSpectrogramLibrosa.__call__), 1: log(mel + eps_log)
};

__global__ __launch_bounds__(256) void logmel_kernel(LogmelArgs a) {
    __shared__ float s_tw[2 * NFFT];
    __shared__ float s_re[4][CPAD];
    __shared__ float s_im[4][CPAD];
    __shared__ float s_pw[4][516];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // melw [n_mels][taps], tile [n_mels][FPB+1]
    float* s_melw = s_dyn;
    float* s_tile = s_dyn + a.n_mels * a.taps;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * FPB;
    for (int i = threadIdx.x; i < 2 * NFFT; i += 256) s_tw[i] = a.tw[i];
    for (int i = threadIdx.x; i < a.n_mels * a.taps; i += 256) s_melw[i] = a.melw[i];
    __syncthreads();

    const float* x = a.wave + (long)b * a.L;
    float* re = s_re[w];
    float* im = s_im[w];
    float* pw = s_pw[w];

    for (int s = 0; s < FPB / 4; ++s) {
        const int f = w + 4 * s;
        const int t = min(t0 + f, a.T - 1);          // clamped duplicate keeps barriers uniform
        C32 v[8];
        // ---- pass 0 (Ns = 1): inputs z[m] = x[2m] + i x[2m+1], m = lane + 64 r
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = 2 * (lane + 64 * r);
            int p0 = t * a.hop + n - NFFT / 2, p1 = p0 + 1;
            p0 = p0 < 0 ? -p0 : (p0 >= a.L ? 2 * (a.L - 1) - p0 : p0);
            p1 = p1 < 0 ? -p1 : (p1 >= a.L ? 2 * (a.L - 1) - p1 : p1);
            v[r].x = x[p0] * a.win[n];
            v[r].y = x[p1] * a.win[n + 1];
        }
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int o = pidx(8 * lane + r); re[o] = v[r].x; im[o] = v[r].y; }
        __syncthreads();
        // ---- pass 1 (Ns = 8): twiddle W_64^{r k} = W_1024^{16 r k}
        {
            const int k = lane & 7;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int o = pidx(lane + 64 * r);
                C32 z = {re[o], im[o]};
                const int ti = 16 * r * k;
                v[r] = cmul(z, C32{s_tw[2 * ti], s_tw[2 * ti + 1]});
            }
            dft8(v);
            __syncthreads();
            const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int o = pidx(j0 + 8 * r); re[o] = v[r].x; im[o] = v[r].y; }
        }
        __syncthreads();
        // ---- pass 2 (Ns = 64): twiddle W_512^{r k} = W_1024^{2 r k}; output Z[lane + 64 r] in natural order
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int o = pidx(lane + 64 * r);
                C32 z = {re[o], im[o]};
                const int ti = 2 * r * lane;
                v[r] = cmul(z, C32{s_tw[2 * ti], s_tw[2 * ti + 1]});
            }
            dft8(v);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int o = pidx(lane + 64 * r); re[o] = v[r].x; im[o] = v[r].y; }
        }
        __syncthreads();
        // ---- real-FFT split: X[k] = E - i W_1024^k O,  E = (Z[k] + conj Z[512-k])/2, O = (Z[k] - conj Z[512-k])/2
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = lane + 64 * r;
            const int o2 = pidx((512 - k) & 511);
            const C32 A = v[r];
            const C32 Bc = {re[o2], -im[o2]};
            const C32 E = {0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y)};
            const C32 O = {0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y)};
            const C32 WO = cmul(C32{s_tw[2 * k], s_tw[2 * k + 1]}, O);
            const C32 X = {E.x + WO.y, E.y - WO.x};                 // E - i*WO
            pw[k] = X.x * X.x + X.y * X.y + a.eps_pow;
            if (k == 0) { const float ny = A.x - A.y; pw[512] = ny * ny + a.eps_pow; }
        }
        __syncthreads();
        // ---- sparse mel rows + log
        for (int m = lane; m < a.n_mels; m += 64) {
            const float* wrow = s_melw + m * a.taps;
            const int st = a.mel_start[m];
            float acc = 0.f;
            for (int i = 0; i < a.taps; ++i) acc += wrow[i] * pw[min(st + i, 512)];
            s_tile[m * (FPB + 1) + f] = a.apply_log ? logf(acc + a.eps_log) : acc;
        }
        __syncthreads();
    }
    // ---- [n_mels][FPB] tile -> out[b][m][t0 + f]
    for (int i = threadIdx.x; i < a.n_mels * FPB; i += 256) {
        const int m = i / FPB, f = i % FPB;
        if (t0 + f < a.T) a.out[((long)b * a.n_mels + m) * a.T + t0 + f] = s_tile[m * (FPB + 1) + f];
    }
}

}  // namespace

extern "C" int audiossl_logmel_fwd(const float* wave, float* out, int B, int L, int T, int n_fft, int hop, int n_mels,
                                   int taps, const float* win, const float* tw, const float* melw,
                                   const int* mel_start, float eps_pow, float eps_log, int apply_log, void* stream) {
    ASSL_REQUIRE(wave && out && win && tw && melw && mel_start);
    ASSL_REQUIRE(n_fft == NFFT);                       // the FFT is specialised for the reference's n_fft=1024
    ASSL_REQUIRE(B > 0 && hop > 0 && L > NFFT / 2 && n_mels > 0 && n_mels <= 128 && taps > 0 && taps <= 513);
    ASSL_REQUIRE(T == 1 + L / hop);
    const size_t dyn = sizeof(float) * ((size_t)n_mels * taps + (size_t)n_mels * (FPB + 1));
    ASSL_REQUIRE(dyn <= 28 * 1024);
    LogmelArgs a{wave, out, B, L, T, hop, n_mels, taps, win, tw, melw, mel_start, eps_pow, eps_log, apply_log};
    hipLaunchKernelGGL(logmel_kernel, dim3(ceil_div(T, FPB), B), dim3(256), dyn, static_cast<hipStream_t>(stream), a);
    ASSL_LAUNCH_CHECK();
}
