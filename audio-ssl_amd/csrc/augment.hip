// K2-K5: running normalisation + two augmented views (log-mixup-exp, random-resize-crop, band masks).
//
// Batched device form of the reference's per-clip CPU chain (`src/augmentations/__init__.py:32-35`,
// `augmentations.py:8-12, 14-61, 82-116, 215-282`).  All randomness stays on the host (the reference
// draws from numpy / python `random`; indices must be bit-exact), so the kernels take parameter tables.
//   clip_moments   : per-clip sum / sum-of-squares in fp64                       (read 4*F*T B/clip)
//   runnorm_scan   : the sequential RunningNorm recurrence over the batch, one thread (B steps)
//   aug_normalize  : (x - mu_c) / sd_c into the device ring that doubles as the mixup memory bank
//   aug_views      : per (clip, view) block: mix with the partner clip into LDS, then bicubic
//                    (A=-0.75, align_corners) resize of the random crop of the zero canvas
//   mask_fill      : SpecAugment band fill (0 or running mean), sequential per clip
// HBM roofline: 155,136 B/clip = x + <=2 partners read, 2 views + bank slot written.
#include "common.h"

namespace {

constexpr float F32_EPS = 1.1920928955078125e-07f;
constexpr float F32_MAX = 3.4028234663852886e+38f;

__global__ __launch_bounds__(256) void clip_moments_kernel(const float* __restrict__ x, double* __restrict__ mom, int n) {
    __shared__ double sh[16];
    const float* p = x + (long)blockIdx.x * n;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double v = p[i]; s += v; q += v * v; }
    s = block_sum(s, sh);
    q = block_sum(q, sh);
    if (threadIdx.x == 0) { mom[2 * blockIdx.x] = s; mom[2 * blockIdx.x + 1] = q; }
}

// (the moments are staged in LDS first: the scan itself is one thread, ~20 cycles per clip)
// state_i = {n_seen, max_update}; state_f = {mu, s2}.  Reference recurrence (augmentations.py:222-227):
// first sample sets, later samples do  v += (new - v) / n  with n the count BEFORE the increment.
__global__ void runnorm_scan_kernel(const double* __restrict__ mom, int B, int n_elem, long long* state_i,
                                    float* state_f, float* __restrict__ mu_out, float* __restrict__ sd_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];           // [2B] moments in, (mu, sd) float pairs out
    for (int i = threadIdx.x; i < 2 * B; i += blockDim.x) sm[i] = mom[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        long long n = state_i[0];
        const long long max_update = state_i[1];
        float mu = state_f[0], s2 = state_f[1];
        const double inv = 1.0 / (double)n_elem;
        for (int c = 0; c < B; ++c) {
            if (n < max_update) {
                const double ex = sm[2 * c] * inv, ex2 = sm[2 * c + 1] * inv;
                const float m = (float)ex;
                mu = (n == 0) ? m : mu + (m - mu) / (float)n;
                const double mud = (double)mu;
                const float v = (float)(ex2 - 2.0 * mud * ex + mud * mud);
                s2 = (n == 0) ? v : s2 + (v - s2) / (float)n;
                ++n;
            }
            float* o = reinterpret_cast<float*>(&sm[2 * c]);
            o[0] = mu;
            o[1] = fminf(fmaxf(sqrtf(s2), F32_EPS), F32_MAX);
        }
        state_i[0] = n;
        state_f[0] = mu;
        state_f[1] = s2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < B; c += blockDim.x) {
        const float* o = reinterpret_cast<const float*>(&sm[2 * c]);
        mu_out[c] = o[0];
        sd_out[c] = o[1];
    }
}

__global__ __launch_bounds__(256) void aug_normalize_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                            const float* __restrict__ sd, float* __restrict__ bank,
                                                            long slot0, int R, int n) {
    const int c = blockIdx.y;
    const float m = mu[c], s = sd[c];
    const float* src = x + (long)c * n;
    float* dst = bank + ((slot0 + c) % R) * (long)n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = (src[i] - m) / s;
}

__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
    const float A = -0.75f;
    float x = t + 1.f;
    c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = x + 1.f;
    c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// ip[(c*2+v)*8 + {0..7}] = {self_slot, partner_slot(-1: none), i, j, h, w, do_rrc, -}
// fp[(c*2+v)*2 + {0,1}]  = {coef_self, coef_partner}  (fp32 roundings of (1-alpha) and 1-(1-alpha))
struct ViewArgs {
    const float* bank; const int* ip; const float* fp; float* out1; float* out2;
    int F, T, canvas_h, canvas_w, y0, x0, log_mix;
};

__global__ __launch_bounds__(256) void aug_views_kernel(ViewArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mixed[];
    const int c = blockIdx.x, view = blockIdx.y;
    const int* ip = a.ip + (c * 2 + view) * 8;
    const float* fp = a.fp + (c * 2 + view) * 2;
    const int n = a.F * a.T;
    const float* xs = a.bank + (long)ip[0] * n;
    const int partner = ip[1];
    if (partner >= 0) {
        const float* zs = a.bank + (long)partner * n;
        const float cs = fp[0], cp = fp[1];
        if (a.log_mix) {
            for (int i = threadIdx.x; i < n; i += 256) mixed[i] = logf(cs * expf(xs[i]) + cp * expf(zs[i]) + F32_EPS);
        } else {
            for (int i = threadIdx.x; i < n; i += 256) mixed[i] = cp * zs[i] + cs * xs[i];
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) mixed[i] = xs[i];
    }
    __syncthreads();
    float* out = (view == 0 ? a.out1 : a.out2) + (long)c * n;
    if (!ip[6]) {
        for (int i = threadIdx.x; i < n; i += 256) out[i] = mixed[i];
        return;
    }
    const int ci = ip[2], cj = ip[3], h = ip[4], w = ip[5];
    const float sy = a.F > 1 ? (float)(h - 1) / (float)(a.F - 1) : 0.f;
    const float sx = a.T > 1 ? (float)(w - 1) / (float)(a.T - 1) : 0.f;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int oy = idx / a.T, ox = idx - oy * a.T;
        const float ry = sy * (float)oy, rx = sx * (float)ox;
        const float fy = floorf(ry), fx = floorf(rx);
        float wy[4], wx[4];
        cubic_coeffs(fminf(fmaxf(ry - fy, 0.f), 1.f), wy);
        cubic_coeffs(fminf(fmaxf(rx - fx, 0.f), 1.f), wx);
        const int iy = (int)fy, ix = (int)fx;
        int xi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xi[q] = cj + min(max(ix - 1 + q, 0), w - 1) - a.x0;     // image column or outside
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int yy = ci + min(max(iy - 1 + p, 0), h - 1) - a.y0;                     // image row or outside
            const bool rowin = yy >= 0 && yy < a.F;
            float row = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = rowin && xi[q] >= 0 && xi[q] < a.T;
                const float s = in ? mixed[yy * a.T + xi[q]] : 0.f;
                row = row + s * wx[q];
            }
            acc = acc + row * wy[p];
        }
        out[idx] = acc;
    }
}

// SpecAugment band fill, `extras/delores-s/specaugment.py:68-122`, on [B][F][T] views in place.
// tab[(c*max_masks+k)*4 + {0..3}] = {axis (0 = time, 1 = freq, -1 = stop), start, end, -}; masks are applied in
// order and the fill value is the mean of the tensor as it stands (earlier masks included) unless zero_fill.
__global__ __launch_bounds__(256) void mask_fill_kernel(float* __restrict__ x, const int* __restrict__ tab, int max_masks,
                                                        int F, int T, int zero_fill) {
    __shared__ double sh[16];
    float* p = x + (long)blockIdx.x * F * T;
    const int n = F * T;
    for (int k = 0; k < max_masks; ++k) {
        const int* e = tab + ((long)blockIdx.x * max_masks + k) * 4;
        const int axis = e[0], st = e[1], en = e[2];
        if (axis < 0) break;                                   // uniform across the block
        float fill = 0.f;
        if (!zero_fill) {
            double s = 0.0;
            for (int i = threadIdx.x; i < n; i += 256) s += (double)p[i];
            s = block_sum(s, sh);
            fill = (float)(s / (double)n);
        }
        __syncthreads();
        if (en > st) {
            if (axis == 1) {           // frequency rows [st, en)
                const int cnt = (en - st) * T;
                for (int i = threadIdx.x; i < cnt; i += 256) p[st * T + i] = fill;
            } else {                   // time columns [st, en)
                const int wdt = en - st, cnt = wdt * F;
                for (int i = threadIdx.x; i < cnt; i += 256) { const int f = i / wdt, t = st + i - f * wdt; p[f * T + t] = fill; }
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int audiossl_clip_moments(const float* x, double* mom, int B, int n, void* stream) {
    ASSL_REQUIRE(x && mom && B > 0 && n > 0);
    hipLaunchKernelGGL(clip_moments_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), x, mom, n);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_runnorm_scan(const double* mom, int B, int n_elem, long long* state_i, float* state_f,
                                     float* mu_out, float* sd_out, void* stream) {
    ASSL_REQUIRE(mom && state_i && state_f && mu_out && sd_out && B > 0 && B <= 3840 && n_elem > 0);
    hipLaunchKernelGGL(runnorm_scan_kernel, dim3(1), dim3(256), sizeof(double) * 2 * B, static_cast<hipStream_t>(stream), mom, B, n_elem,
                       state_i, state_f, mu_out, sd_out);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_aug_normalize(const float* x, const float* mu, const float* sd, float* bank, long slot0,
                                      int R, int B, int n, void* stream) {
    ASSL_REQUIRE(x && mu && sd && bank && R > 0 && B > 0 && B <= R && n > 0 && slot0 >= 0);
    hipLaunchKernelGGL(aug_normalize_kernel, dim3(ceil_div(n, 1024), B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, mu, sd, bank, slot0, R, n);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_aug_views(const float* bank, int R, const int* ip, const float* fp, float* out1, float* out2,
                                  int B, int F, int T, int canvas_h, int canvas_w, int log_mix, void* stream) {
    ASSL_REQUIRE(bank && ip && fp && out1 && out2 && B > 0 && F > 0 && T > 0 && R > 0);
    ASSL_REQUIRE(canvas_h >= F && canvas_w >= T && (size_t)F * T * 4 <= 60 * 1024);
    ViewArgs a{bank, ip, fp, out1, out2, F, T, canvas_h, canvas_w, (canvas_h - F) / 2, (canvas_w - T) / 2, log_mix};
    hipLaunchKernelGGL(aug_views_kernel, dim3(B, 2), dim3(256), (size_t)F * T * 4, static_cast<hipStream_t>(stream), a);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_mask_fill(float* x, const int* tab, int n_img, int max_masks, int F, int T, int zero_fill,
                                  void* stream) {
    ASSL_REQUIRE(x && tab && n_img > 0 && max_masks > 0 && F > 0 && T > 0);
    hipLaunchKernelGGL(mask_fill_kernel, dim3(n_img), dim3(256), 0, static_cast<hipStream_t>(stream), x, tab, max_masks,
                       F, T, zero_fill);
    ASSL_LAUNCH_CHECK();
}
