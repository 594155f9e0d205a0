// K2-K5: running normalisation + two augmented views (log-mixup-exp, random-resize-crop, band masks).
//
// Batched device form of the reference's per-clip CPU chain (`src/augmentations/__init__.py:32-35`,
// `augmentations.py:8-12, 14-61, 82-116, 215-282`).  All randomness stays on the host (the reference
// draws from numpy / python `random`; indices must be bit-exact), so the kernels take parameter tables.
//   clip_moments   : per-clip sum / sum-of-squares in fp64                       (read 4*F*T B/clip)
//   runnorm_scan   : the sequential RunningNorm recurrence over the batch, bit for bit, as two pipelined serial chains
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

// state_i = {n_seen, max_update}; state_f = {mu, s2}.  Reference recurrence (augmentations.py:222-227), fp32, clip after clip:
// first sample sets, later samples do  v += (new - v) / n  with n the count BEFORE the increment; the variance sample of clip
// c uses the ALREADY updated mean.  The recurrence is kept bit for bit (a re-associated parallel scan drifts from the fp32
// evaluation by up to ~30 ulp over an epoch), but it is taken apart so that only two short dependent chains stay serial:
//   * every 1/n is computed ahead, in parallel, in fp64.  (float)((double)d * fl64(1/n)) equals the correctly rounded fp32
//     quotient d / (float)n for n < 2^24: the exact quotient of a 24-bit by a <=24-bit number is at least 2^-50 (relative)
//     away from every fp32 rounding boundary, the fp64 product is within 2^-52 of it.  Larger n take the true division.
//   * wave 0 runs the mean chain (sub, cvt, mul, cvt, add per clip) 64 clips ahead of wave 1, which computes the 64 variance
//     samples of the previous block in parallel across its lanes and then runs the variance chain.
// 512 clips: 133 us as one thread doing everything -> ~20 us.
// 64 steps of  v <- v + (float)((double)(x_c - v) * r_c)  executed by the whole wave on wave-uniform operands: lane c holds
// (x_c, r_c), each step broadcasts its pair with v_readlane (scalar registers), so the dependent chain is five VALU
// instructions per step with no LDS access in it; lane c keeps the state after step c.
__device__ __forceinline__ float chain64(float v, float x_mine, double r_mine, float& keep) {
    const int rlo = __double2loint(r_mine), rhi = __double2hiint(r_mine);
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x_mine), c));
        const double r = __hiloint2double(__builtin_amdgcn_readlane(rhi, c), __builtin_amdgcn_readlane(rlo, c));
        v = v + (float)((double)(x - v) * r);
        keep = ((int)(threadIdx.x & 63) == c) ? v : keep;
    }
    return v;
}

__global__ __launch_bounds__(128) void runnorm_scan_kernel(const double* __restrict__ mom, int B, int n_elem, long long* state_i,
                                                           float* state_f, float* __restrict__ mu_out, float* __restrict__ sd_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* rn = sm;                                        // [B] 1 / (float)n
    float* m_s = reinterpret_cast<float*>(rn + B);          // [B] clip means
    float* mu_s = m_s + B;                                  // [B] running mean after clip c
    float* v_s = mu_s + B;                                  // [B] variance sample of clip c
    float* s2_s = v_s + B;                                  // [B] running variance after clip c
    const long long n0 = state_i[0], max_update = state_i[1];
    const double inv = 1.0 / (double)n_elem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // rn[c] = 0 for clips that do not update (n >= max_update) and for the very first sample (n == 0, which SETS the state):
    // the chain step  v + (float)((double)(new - v) * rn)  then leaves v untouched, so the serial loops are branch-free
    const bool big = n0 + B > (1 << 24);                    // beyond 2^24 samples the reciprocal is not exact: true division
    for (int c = tid; c < B; c += 128) {
        m_s[c] = (float)(mom[2 * c] * inv);
        const long long n = n0 + c;
        rn[c] = (n > 0 && n < max_update) ? (big ? (double)(float)n : 1.0 / (double)(float)n) : 0.0;
    }
    float mu = state_f[0], s2 = state_f[1];
    const bool first = n0 == 0 && max_update > 0;            // clip 0 is the first sample ever
    __syncthreads();
    const int nblk = (B + 63) / 64;
    for (int k = 0; k <= nblk; ++k) {
        if (wave == 0 && k < nblk) {
            const int c0 = 64 * k, c = c0 + lane;
            if (!big) {
                if (k == 0 && first) mu = m_s[0];                        // rn[0] = 0: the chain leaves it in place
                float keep = 0.f;
                mu = chain64(mu, c < B ? m_s[c] : 0.f, c < B ? rn[c] : 0.0, keep);
                if (c < B) mu_s[c] = keep;
            } else if (lane == 0) {
                int i = c0;
                const int c1 = min(c0 + 64, B);
                if (i == 0 && first) { mu = m_s[0]; mu_s[0] = mu; i = 1; }
                for (; i < c1; ++i) { if (rn[i] != 0.0) mu = mu + (m_s[i] - mu) / (float)rn[i]; mu_s[i] = mu; }
            }
        }
        if (wave == 1 && k >= 1) {
            const int c0 = 64 * (k - 1), c1 = min(c0 + 64, B), c = c0 + lane;
            float v = 0.f;
            if (c < c1) {
                const double ex = mom[2 * c] * inv, ex2 = mom[2 * c + 1] * inv, mud = (double)mu_s[c];
                v = (float)(ex2 - 2.0 * mud * ex + mud * mud);
            }
            if (!big) {
                if (k == 1 && first) s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
                float keep = 0.f;
                s2 = chain64(s2, v, c < c1 ? rn[c] : 0.0, keep);
                if (c < c1) s2_s[c] = keep;
            } else {
                if (c < c1) v_s[c] = v;
                __threadfence_block();                      // the lanes' v_s stores before lane 0 reads them
                if (lane == 0) {
                    int i = c0;
                    if (i == 0 && first) { s2 = v_s[0]; s2_s[0] = s2; i = 1; }
                    for (; i < c1; ++i) { if (rn[i] != 0.0) s2 = s2 + (v_s[i] - s2) / (float)rn[i]; s2_s[i] = s2; }
                }
            }
        }
        __syncthreads();
    }
    for (int c = tid; c < B; c += 128) {
        mu_out[c] = mu_s[c];
        sd_out[c] = fminf(fmaxf(sqrtf(s2_s[c]), F32_EPS), F32_MAX);
    }
    if (tid == 0) {
        long long upd = max_update - n0;
        upd = upd < 0 ? 0 : (upd > B ? B : upd);
        state_i[0] = n0 + upd;
        state_f[0] = mu_s[B - 1];
        state_f[1] = s2_s[B - 1];
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

// ---- Kmix (`src/augmentations/augmentations.py:119-189`): cluster-guided mixup applied to the finished views.
// kmix_cluster: cluster[v] = argmin_c || mean_T(view_v) - centroid_c ||  (centroids pre-normalised to unit rows, as get_index
//   does; for unit rows the nearest centroid of the raw mean and of the normalised mean coincide, so one id serves both the
//   `point_cluster` of a view and its `memory_centroid_dist` once it is in the bank).  First minimum wins (torch.argmin).
// kmix_apply : out_v = log(cs e^x + cp e^z + eps) (or cs x + cp z), z = ring slot plan[v]; slot < 0: out = x.
__global__ __launch_bounds__(256) void kmix_cluster_kernel(const float* __restrict__ views, const float* __restrict__ cent, int F, int T,
                                                           int K, int* __restrict__ cluster) {
    __shared__ float m[128];
    __shared__ float best_d[256];
    __shared__ int best_i[256];
    const float* x = views + (long)blockIdx.x * F * T;
    for (int f = threadIdx.x >> 2; f < F; f += 64) {                 // four lanes per mel row
        float s = 0.f;
        for (int t = threadIdx.x & 3; t < T; t += 4) s += x[f * T + t];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if ((threadIdx.x & 3) == 0) m[f] = s / (float)T;
    }
    __syncthreads();
    float bd = 3.0e38f;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < K; c += 256) {
        float d = 0.f;
        for (int f = 0; f < F; ++f) { const float e = m[f] - cent[c * F + f]; d += e * e; }
        if (d < bd) { bd = d; bi = c; }
    }
    best_d[threadIdx.x] = bd; best_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float d2 = best_d[threadIdx.x + o];
            const int i2 = best_i[threadIdx.x + o];
            if (d2 < best_d[threadIdx.x] || (d2 == best_d[threadIdx.x] && i2 < best_i[threadIdx.x])) { best_d[threadIdx.x] = d2; best_i[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cluster[blockIdx.x] = best_i[0];
}

__global__ __launch_bounds__(256) void kmix_apply_kernel(const float* __restrict__ views, const float* __restrict__ ring,
                                                         const int* __restrict__ slot, const float* __restrict__ coef, int n,
                                                         int log_mix, float* __restrict__ out) {
    const int v = blockIdx.y;
    const int z = slot[v];
    const float cs = coef[2 * v], cp = coef[2 * v + 1];
    const float* x = views + (long)v * n;
    const float* zs = ring + (long)(z < 0 ? 0 : z) * n;
    float* o = out + (long)v * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float xv = x[i];
        if (z < 0) o[i] = xv;
        else o[i] = log_mix ? logf(cs * expf(xv) + cp * expf(zs[i]) + F32_EPS) : cp * zs[i] + cs * xv;
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
    const size_t dyn = (size_t)B * (sizeof(double) + 4 * sizeof(float));             // 24 B per clip: 92 KB at the 3,840-clip limit
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(runnorm_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                96 * 1024) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(runnorm_scan_kernel, dim3(1), dim3(128), dyn, static_cast<hipStream_t>(stream), mom, B, n_elem,
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

extern "C" int audiossl_kmix_cluster(const float* views, const float* centroids, int n_views, int F, int T, int K, int* cluster,
                                     void* stream) {
    ASSL_REQUIRE(views && centroids && cluster && n_views > 0 && F > 0 && F <= 128 && T > 0 && K > 0);
    hipLaunchKernelGGL(kmix_cluster_kernel, dim3(n_views), dim3(256), 0, static_cast<hipStream_t>(stream), views, centroids, F, T, K,
                       cluster);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_kmix_apply(const float* views, const float* ring, const int* slot, const float* coef, int n_views, int n,
                                   int log_mix, float* out, void* stream) {
    ASSL_REQUIRE(views && ring && slot && coef && out && n_views > 0 && n > 0);
    hipLaunchKernelGGL(kmix_apply_kernel, dim3(ceil_div(n, 1024), n_views), dim3(256), 0, static_cast<hipStream_t>(stream), views,
                       ring, slot, coef, n, log_mix, out);
    ASSL_LAUNCH_CHECK();
}
