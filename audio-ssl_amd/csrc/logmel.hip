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
#include <cstdlib>
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

// ---------------------------------------------------------------------------------------------------------------------
// Second form: one PERSISTENT wave per run of frames, nothing but the samples and the results crosses HBM more than once.
// The kernel above is bound by its LDS traffic and scalar fp32 issue, not by memory (3 % of the HBM roof): every frame fetched
// 12 KB of twiddles and up to 23 KB of padded mel rows from LDS, its window from global memory, and used one fp32 lane-op per
// real multiply-add.  Here
//   * window, the three twiddle sets and the wave's mel rows live in registers for the whole launch (a lane is a mel row);
//   * complex values travel as float2 (packed fp32 math, 8-byte LDS accesses); the conjugate-mirror operand of the real-FFT
//     split comes from ds_bpermute instead of a third LDS exchange; no workgroup barrier anywhere (waves are independent);
//   * the mel row is read as 16-byte aligned LDS chunks against register weights (13 ds_read_b128 instead of 90 ds_read_b32);
//   * interior frames load their 1,024 samples as eight coalesced float2 per lane (no reflect arithmetic).
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 cmulv(v2 a, v2 b) { return v2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2 mulmi(v2 a) { return v2{a.y, -a.x}; }     // * (-i)
__device__ __forceinline__ void dft8v(v2* v) {
    const float s = 0.70710678118654752440f;
    v2 a0 = v[0] + v[4], a1 = v[0] - v[4];
    v2 a2 = v[2] + v[6], a3 = mulmi(v[2] - v[6]);
    v2 a4 = v[1] + v[5], a5 = v[1] - v[5];
    v2 a6 = v[3] + v[7], a7 = mulmi(v[3] - v[7]);
    v2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
    v2 b4 = a4 + a6, b6 = mulmi(a4 - a6);
    v2 t5 = a5 + a7, t7 = a5 - a7;
    v2 b5 = v2{(t5.x + t5.y) * s, (t5.y - t5.x) * s};          // * e^{-i pi/4}
    v2 b7 = v2{(t7.y - t7.x) * s, (-t7.x - t7.y) * s};         // * e^{-3i pi/4}
    v[0] = b0 + b4; v[4] = b0 - b4;
    v[1] = b1 + b5; v[5] = b1 - b5;
    v[2] = b2 + b6; v[6] = b2 - b6;
    v[3] = b3 + b7; v[7] = b3 - b7;
}
__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }     // LDS is in-order per wave: only the compiler must not reorder

constexpr int EXW = 576;                // float2 exchange buffer of a wave (512 + 512 / 8 padding)
constexpr int PWW = 576;                // floats: 513 power bins + zero tail up to the last aligned mel chunk
// Output staging: a wave collects up to OTF consecutive frames of its run in an LDS tile [mel row][frame] and writes them out as
// row segments of up to OTF * 4 = 128 contiguous bytes (32 lanes per mel row, 2 rows per store instruction).  Storing each frame on its
// own - one float per lane, lane = mel row, 4 * T bytes apart - made every store instruction 64 partial-line writes: 140 MB left
// the L2 per launch for 13.2 MB of output (PMC WRITE_SIZE, round 2).
constexpr int OTF = 32, OTP = OTF + 1;  // frames per tile; row pitch in floats (odd: the per-frame column writes are conflict-free)
// (OTF = 16: 64-byte segments at the 404-byte row pitch of T = 101 straddle three 32-byte sectors - 23.1 MB left the L2 for 13.2 MB
// of output; at B = 512 a wave's run is 25-26 frames of one clip, so 32 columns give ONE flush of ~100-byte segments per run)

template <int MPL, int NCH>
__global__ __launch_bounds__(256) void logmel2_kernel(LogmelArgs a, int total_frames) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2* ex = reinterpret_cast<v2*>(s_dyn + wave * (2 * EXW + PWW + 64 * MPL * OTP));
    float* pw = reinterpret_cast<float*>(ex + EXW);
    float* ot = pw + PWW;                                    // [64 * MPL][OTP] output tile of this wave
    for (int i = 513 + lane; i < PWW; i += 64) pw[i] = 0.f;

    // ---- per-lane constants
    v2 win[8], tw1[8], tw2[8], tws[8];
    const v2* twt = reinterpret_cast<const v2*>(a.tw);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = 2 * (lane + 64 * r);
        win[r] = v2{a.win[n], a.win[n + 1]};
        tw1[r] = twt[16 * r * (lane & 7)];
        tw2[r] = twt[2 * r * lane];
        tws[r] = twt[lane + 64 * r];
    }
    float mw[MPL][4 * NCH];
    int ma[MPL];
#pragma unroll
    for (int q = 0; q < MPL; ++q) {
        const int m = lane + 64 * q;
        const int st = m < a.n_mels ? a.mel_start[m] : 0;
        ma[q] = st & ~3;
#pragma unroll
        for (int j = 0; j < 4 * NCH; ++j) {
            const int i = ma[q] + j - st;
            mw[q][j] = (m < a.n_mels && i >= 0 && i < a.taps) ? a.melw[m * a.taps + i] : 0.f;
        }
    }
    const int mirror = ((64 - lane) & 63) * 4;               // ds_bpermute source lane (bytes)
    const int j0 = (lane >> 3) * 64 + (lane & 7);

    // frame range of this wave - wave-uniform values kept in scalar registers (readfirstlane), so the frame loop, the
    // clip / frame split and the interior-or-border choice are scalar code
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nw = gridDim.x * 4;
    const int f0 = (int)((long)gw * total_frames / nw), f1 = (int)((long)(gw + 1) * total_frames / nw);
    // raw samples of frame f: z[m] = x[2m] + i x[2m+1], m = lane + 64 r (reflected at the clip borders)
    auto load_frame = [&](int f, v2 (&z)[8]) {
        const int b = f / a.T, t = f - b * a.T;
        const float* x = a.wave + (long)b * a.L;
        const int base = t * a.hop - NFFT / 2;
        if (base >= 0 && base + NFFT <= a.L) {               // interior frame: eight coalesced float2 per lane
            const v2* src = reinterpret_cast<const v2*>(x + base);
#pragma unroll
            for (int r = 0; r < 8; ++r) z[r] = src[lane + 64 * r];
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int p0 = base + 2 * (lane + 64 * r), p1 = p0 + 1;
                p0 = p0 < 0 ? -p0 : (p0 >= a.L ? 2 * (a.L - 1) - p0 : p0);
                p1 = p1 < 0 ? -p1 : (p1 >= a.L ? 2 * (a.L - 1) - p1 : p1);
                z[r] = v2{x[p0], x[p1]};
            }
        }
    };
    v2 nxt[8];
    if (f0 < f1) load_frame(f0, nxt);
    int tl = 0;                                              // column of the output tile the current frame goes to (wave-uniform)
    for (int f = f0; f < f1; ++f) {
        const int b = f / a.T, t = f - b * a.T;
        v2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = nxt[r] * win[r];
        if (f + 1 < f1) load_frame(f + 1, nxt);              // the next frame's samples travel while this frame is transformed
        // ---- pass 0 (Ns = 1)
        dft8v(v);
        wave_fence();
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[pidx(8 * lane + r)] = v[r];
        wave_fence();
        // ---- pass 1 (Ns = 8)
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ex[pidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulv(v[r], tw1[r]);
        dft8v(v);
        wave_fence();
#pragma unroll
        for (int r = 0; r < 8; ++r) ex[pidx(j0 + 8 * r)] = v[r];
        wave_fence();
        // ---- pass 2 (Ns = 64): Z[lane + 64 r]
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ex[pidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulv(v[r], tw2[r]);
        dft8v(v);
        // ---- real-FFT split; Z[512 - k] sits in lane 64 - lane, register 7 - r (lane 0: its own register 8 - r)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const v2 own = v[(8 - r) & 7], oth = v[7 - r];
            v2 zm;
            zm.x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(oth.x)));
            zm.y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(oth.y)));
            if (lane == 0) zm = own;
            const v2 A = v[r], Bc = v2{zm.x, -zm.y};
            const v2 E = (A + Bc) * 0.5f, O = (A - Bc) * 0.5f;
            const v2 WO = cmulv(tws[r], O);
            const v2 X = v2{E.x + WO.y, E.y - WO.x};
            pw[lane + 64 * r] = X.x * X.x + X.y * X.y + a.eps_pow;
            if (r == 0 && lane == 0) { const float ny = A.x - A.y; pw[512] = ny * ny + a.eps_pow; }
        }
        wave_fence();
        // ---- mel rows: aligned 16-byte chunks of the power spectrum against register weights, then log
#pragma unroll
        for (int q = 0; q < MPL; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(pw + ma[q] + 4 * c);
                acc += mw[q][4 * c] * p[0] + mw[q][4 * c + 1] * p[1] + mw[q][4 * c + 2] * p[2] + mw[q][4 * c + 3] * p[3];
            }
            ot[(lane + 64 * q) * OTP + tl] = a.apply_log ? logf(acc + a.eps_log) : acc;
        }
        wave_fence();
        // ---- flush the tile when it is full, at the end of a clip (the next frame belongs to another clip's rows) and at the
        //      end of the run: frames t - tl .. t of clip b, OTF lanes per mel row
        if (tl == OTF - 1 || t == a.T - 1 || f == f1 - 1) {
            const int col = lane & (OTF - 1), t0 = t - tl;
            float* const dst = a.out + (long)b * a.n_mels * a.T + t0 + col;
            constexpr int RPI = 64 / OTF;                    // mel rows per store instruction
#pragma unroll 4
            for (int i = 0; i < 64 * MPL / RPI; ++i) {
                const int row = RPI * i + lane / OTF;
                const float v = ot[row * OTP + col];
                if (col <= tl && row < a.n_mels) dst[(long)row * a.T] = v;
            }
            wave_fence();
            tl = 0;
        } else {
            ++tl;
        }
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
    // the persistent form needs 8-byte aligned frame starts and mel rows that fit its register budget
    static const int v1_only = getenv("AUDIOSSL_LOGMEL_V1") ? atoi(getenv("AUDIOSSL_LOGMEL_V1")) : 0;
    const bool v2_ok = !v1_only && hop % 2 == 0 && L % 2 == 0 && (((uintptr_t)wave) & 7) == 0 && L >= NFFT;
    if (v2_ok && (n_mels <= 64 ? taps + 3 <= 52 : taps + 3 <= 28)) {
        const long total = (long)B * T;
        const int blocks = (int)min((long)512, (total + 3) / 4);          // two workgroups per CU, ~25 frames per wave at B = 512
        const size_t lds = sizeof(float) * 4 * (2 * EXW + PWW + 64 * (n_mels <= 64 ? 1 : 2) * OTP);
        if (n_mels <= 64) hipLaunchKernelGGL((logmel2_kernel<1, 13>), dim3(blocks), dim3(256), lds, static_cast<hipStream_t>(stream), a, (int)total);
        else              hipLaunchKernelGGL((logmel2_kernel<2, 7>), dim3(blocks), dim3(256), lds, static_cast<hipStream_t>(stream), a, (int)total);
        ASSL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(logmel_kernel, dim3(ceil_div(T, FPB), B), dim3(256), dyn, static_cast<hipStream_t>(stream), a);
    ASSL_LAUNCH_CHECK();
}
