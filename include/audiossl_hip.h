/* audiossl_hip.h - C ABI of libaudiossl_hip.so (AMD MI355X / gfx950 only).
 *
 * The reference (Sreyan88/audio-ssl) has no native layer: its hot path is ATen ops called from
 * Python.  This header is therefore the boundary a reference maintainer would bind with ctypes
 * (see INTEGRATION.md); each entry point names the reference code it replaces (file:line under
 * the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (AUDIOSSL_E*); nothing throws;
 *   - all pointers are DEVICE pointers owned by the caller (no allocation, no ownership transfer),
 *     16-byte aligned, contiguous unless a leading dimension is given;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous;
 *   - `dtype` selects the storage/MFMA type of activations: 0 = fp32 (exact f32 MFMA, validation /
 *     high-precision path), 1 = bf16 (fp32 accumulate; master weights, statistics and losses stay fp32);
 *   - activations of the conv encoder are channels-last [N][T][F][64] (time, mel, channel), i.e. the
 *     reference's x.permute(0,3,2,1) (src/encoder/audiontt.py:76,83,90,95), so its feature index d*64+c
 *     is contiguous;
 *   - gradient outputs named d* are ACCUMULATED (+=) into caller-zeroed buffers;
 *   - `gdtype` (backward kernels of BatchNorm / pooling): storage type of the INCOMING gradient tensor - 0 = fp32 even
 *     when `dtype` is bf16 (it is a GEMM output, so fp32 is free, and BatchNorm's backward cancels most of it); the
 *     outgoing gradient is stored in `dtype` because it is only ever an MFMA operand;
 *   - `adtype` / `ydtype`: storage type of a BatchNorm INPUT (projector pre-activations, conv outputs) - 0 = fp32 also on
 *     the bf16 path: pre-normalisation values can have |mean| >> std, which 8 significand bits cannot carry.
 */
#ifndef AUDIOSSL_HIP_H
#define AUDIOSSL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUDIOSSL_OK 0
#define AUDIOSSL_EINVAL (-1)   /* bad shape / null pointer / unsupported configuration */
#define AUDIOSSL_ELAUNCH (-2)  /* HIP reported an error at launch */
#define AUDIOSSL_EALIGN (-3)   /* pointer / leading dimension not 16-byte aligned */

/* ---- K1 front end: src/utils/utils.py:20-28 (MelSpectrogramLibrosa.__call__), :43-49 (log) ------------
 * wave [B][L] f32 -> out [B][n_mels][T] f32, T = 1 + L/hop, centre reflect pad, periodic Hann `win`[1024],
 * `tw` = exp(-2 pi i k/1024) as [1024][2], `melw` [n_mels][taps] = packed non-zero run of each filterbank
 * row starting at bin mel_start[m]; out = mel(|X|^2 + eps_pow), or log(. + eps_log) when apply_log.  n_fft must be 1024. */
int audiossl_logmel_fwd(const float* wave, float* out, int B, int L, int T, int n_fft, int hop, int n_mels, int taps,
                        const float* win, const float* tw, const float* melw, const int* mel_start, float eps_pow,
                        float eps_log, int apply_log, void* stream);

/* ---- K2 RunningNorm: src/augmentations/augmentations.py:215-282 ----------------------------------------
 * clip_moments: mom[c] = {sum x, sum x^2} (fp64) of clip c over n = F*T elements.
 * runnorm_scan: sequential recurrence over the batch; state_i = {n_seen, max_update}, state_f = {mu, s2};
 *               writes the (mean, std) each clip is normalised with. */
int audiossl_clip_moments(const float* x, double* mom, int B, int n, void* stream);
int audiossl_runnorm_scan(const double* mom, int B, int n_elem, long long* state_i, float* state_f, float* mu_out,
                          float* sd_out, void* stream);
/* bank[(slot0 + c) % R][:] = (x[c] - mu[c]) / sd[c]; the ring is also the MixupBYOLA memory bank. */
int audiossl_aug_normalize(const float* x, const float* mu, const float* sd, float* bank, long slot0, int R, int B, int n,
                           void* stream);
/* ---- K3+K4 MixupBYOLA + RandomResizeCrop: augmentations.py:8-12, 97-111 and :40-55 ---------------------
 * ip [B][2][8] = {self_slot, partner_slot | -1, i, j, h, w, do_rrc, 0}; fp [B][2][2] = {coef_self, coef_partner}.
 * out1/out2 [B][F][T] f32 = the two views. */
int audiossl_aug_views(const float* bank, int R, const int* ip, const float* fp, float* out1, float* out2, int B, int F,
                       int T, int canvas_h, int canvas_w, int log_mix, void* stream);
/* ---- host planner (CPU function, HOST pointers, no stream): the reference's per-clip random draws, batched ---------
 * Continues numpy's legacy global MT19937 (np_key[624], *np_pos) and python `random`'s MT19937 (py_key, *py_pos) in
 * the reference's draw order (augmentations.py:32-37, 99-102; specaugment.py:80-88) and fills the aug_views /
 * mask_fill tables.  *n_entries = entries appended so far to MixupBYOLA's virtual FIFO (updated).  lens/unit/starts
 * (optional): the random window crop of src/utils/utils.py:166-182, drawn per clip ahead of that clip's views. */
int audiossl_aug_plan_host(uint32_t* np_key, int* np_pos, uint32_t* py_key, int* py_pos, int B, int F, int T,
                           long long clips_seen, long long* n_entries, int R, int n_memory, int use_mix, double ratio,
                           int use_rrc, double fs_lo, double fs_hi, double ts_lo, double ts_hi, int canvas_h,
                           int canvas_w, int use_spec, int spec_F, int spec_T, int spec_nf, int spec_nt, const int* lens,
                           int unit, int* starts, int* ip, float* fp, int* masks);

/* ---- K5 SpecAugment band masks: extras/delores-s/specaugment.py:68-122 ---------------------------------
 * tab [n_img][max_masks][4] = {axis (0 time, 1 freq, -1 stop), start, end, 0}; in place on x [n_img][F][T]. */
/* ---- f2 audio ingest: the resampling of `librosa.core.load(path, sr=16000)` (src/dataset/upstream_dataset.py:55; librosa 0.8.1
 * -> resampy 0.2.2, filter kaiser_best).  y[c][t] = sum_i (win[off_l[t] + i s] + eta_l[t] delta[...]) x[c][n[t] - i]
 *                                              + sum_k (win[off_r[t] + k s] + eta_r[t] delta[...]) x[c][n[t] + 1 + k],  s = index_step,
 * wings cut at the clip ends and the table end, products in fp64 added into an fp32 running sum in resampy's order.  The
 * position tables come from the host (src/dataset/ingest.py), which keeps resampy's float64 time register. */
int audiossl_resample_sinc(const float* x, float* y, int clips, int n_orig, int n_out, const int* n, const int* off_l,
                           const int* off_r, const double* eta_l, const double* eta_r, const double* win, const double* delta,
                           int nwin, int index_step, void* stream);

/* ---- a10 Kmix: src/augmentations/augmentations.py:119-189 (cluster-guided mixup of the finished views) -------------------
 * kmix_cluster: cluster[v] = first argmin_c ||mean_T(views[v]) - centroids[c]|| (centroids [K][F] with unit rows: what
 *   `get_index` computes for the view itself and, once it sits in the memory bank, for the bank entry).
 * kmix_apply: out[v] = log(coef[v][0] e^x + coef[v][1] e^z + eps) with z = ring[slot[v]] (log_mix 0: linear mix; slot < 0:
 *   copy).  Partner choice, the alpha / index draws and the FIFO live on the host (they are numpy-stream exact). */
int audiossl_kmix_cluster(const float* views, const float* centroids, int n_views, int F, int T, int K, int* cluster,
                          void* stream);
int audiossl_kmix_apply(const float* views, const float* ring, const int* slot, const float* coef, int n_views, int n,
                        int log_mix, float* out, void* stream);
int audiossl_mask_fill(float* x, const int* tab, int n_img, int max_masks, int F, int T, int zero_fill, void* stream);

/* ---- K6 stem: src/encoder/audiontt.py:46-50 (features_1) ------------------------------------------------
 * conv1_stats: batch statistics of Conv2d(1,64,3,p=1) output from 54 tap moments (`mom`: 16 x 54 doubles of scratch, the
 *   totals are left in mom[0..53] for the backward); updates running stats; writes scale = gamma*rstd, shift = beta - mean*scale, mean, rstd.
 * conv1_fwd : img [N][F][T] f32 -> pooled [N][T/2][F/2][64] (dtype), conv recomputed, nothing else stored.
 * conv1_bwd : dP (+ dxl [N][F/2*64] / (T/2), may be NULL) -> dW [64][9], dgamma, dbeta (dbias == 0); acc = 32*704
 *             floats of scratch.
 * dtype 0: fp32 VALU convolution, fp32 output (validation path); dtype 2 (forward only): the same convolution, bf16 output
 * (`bf16_hp`).  dtype 1 (forward) / conv_dtype 1 (backward, fp32 gradients): the conv
 * is a bf16 MFMA (taps x weights, K = 9 padded to 16) and so are the nine tap sums of the weight gradient; forward and
 * backward must use the same flavour so the pooling arg-max agrees. */
int audiossl_conv1_stats(const float* img, int N, int F, int T, const float* w, const float* bias, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         double* mom, float* scale, float* shift, float* save_mean, float* save_rstd, void* stream);
/* xl (optional, dtype 1 only): fp32 [AUDIOSSL_CONV1_XL_PARTS][N][F/2 * 64]: four partial sums of the mean over pooled time of
 * `out` (the layer output x_1, audiontt.py:76-78), left by the same launch; tmean3_fwd(x1_parts) adds them in a fixed order -
 * the separate tmean pass re-read the whole pooled map */
#define AUDIOSSL_CONV1_XL_PARTS 4
int audiossl_conv1_fwd(int dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                       const float* scale, const float* shift, void* out, float* xl, void* stream);
/* conv1_bwd: dtype = type of the pooled gradient dP (0 fp32, 1 bf16); with conv_dtype 1 (MFMA recompute) dxl stays fp32 whatever
 * dtype says, with conv_dtype 0 dP and dxl share it */
int audiossl_conv1_bwd(int dtype, int conv_dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                       const float* gamma, const float* scale, const float* shift, const float* mean, const float* rstd,
                       const double* mom, const void* dP, const void* dxl, float* acc, float* dW, float* dbias,
                       float* dgamma, float* dbeta, void* stream);

/* ---- K7/K8 conv blocks 2,3: audiontt.py:52-60, 76-93 ----------------------------------------------------
 * colstats : per-column sum / sum of squares of x [G][M][C] (ld) in fp64 -> [G][C].  `groups` = independent batches that
 *            go through the same layer (the two views of the projector); C % 8 == 0.
 * bn_finalize: train-mode BatchNorm statistics -> per-group scale/shift/mean/rstd [G][C]; running stats (momentum 0.1)
 *            are updated group after group, as the reference's successive module calls do.
 * bn_relu_pool_fwd: Y [N][Ti][Fi][64] -> P [N][Ti/2][Fi/2][64].   tmean_fwd: P -> xl [N][Fo*64] (x_1/x_2/x_3).
 * bn_relu_pool_bwd: dP (+dxl/To) -> dY at every position, dgamma, dbeta (stat = 33*128 floats scratch).
 * im2col3x3 / pack_conv_w / unpack_conv_dw: implicit-GEMM plumbing, tap = kh*3+kw, kh on mel, kw on time. */
int audiossl_colstats(int dtype, const void* x, int groups, long M, int C, long ld, int want_sq, double* sum,
                      double* sumsq, void* stream);
int audiossl_bn_finalize(const double* sum, const double* sumsq, int groups, double count, int C, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* scale, float* shift, float* save_mean, float* save_rstd, void* stream);
int audiossl_bn_relu_pool_fwd(int dtype, int ydtype, const void* Y, const float* scale, const float* shift, void* P, int N,
                              int Ti, int Fi, void* stream);
/* bn_finalize + bn_relu_pool_fwd of one train-mode conv block (64 channels) in one launch: every workgroup derives scale / shift
 * from the fp64 sums itself; scale / shift / mean / rstd (for the backward) and the running buffers are written by workgroup 0. */
int audiossl_bn_relu_pool_train_fwd(int dtype, int ydtype, const void* Y, const double* sum, const double* sumsq, int stat_replicas,
                                    double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, void* P, float* scale, float* shift, float* save_mean, float* save_rstd,
                                    int N, int Ti, int Fi, void* stream);
int audiossl_tmean_fwd(int dtype, int out_f32, const void* P, void* xl, int N, int To, int Fo, void* stream);
/* x_1, x_2, x_3 of one encoder pass (`audiontt.py:76-93`) in one launch; with P1 == NULL and x1_parts given (out_f32 only)
 * x_1 = sum of the AUDIOSSL_CONV1_XL_PARTS parts conv1_fwd left */
int audiossl_tmean3_fwd(int dtype, int out_f32, const void* P1, void* x1, const float* x1_parts, int To1, int Fo1, const void* P2,
                        void* x2, int To2, int Fo2, const void* P3, void* x3, int To3, int Fo3, int N, void* stream);
int audiossl_bn_relu_pool_bwd(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl, const float* scale,
                              const float* shift, const float* mean, const float* rstd, float* stat, void* dY,
                              float* dgamma, float* dbeta, int N, int Ti, int Fi, void* stream);
/* The same with the statistics pass (dbeta, dgamma) taken from the block's POOLED forward output P [N][Ti/2][Fi/2][64] (activation
 * dtype) instead of a first sweep over Y: the routed gradient lives where P > 0, and there xhat = ((P - shift) / scale - mean) rstd. */
int audiossl_bn_relu_pool_bwd_p(int dtype, int ydtype, int gdtype, const void* Y, const void* P, const void* dP, const void* dxl,
                                const float* scale, const float* shift, const float* mean, const float* rstd, float* stat, void* dY,
                                float* dgamma, float* dbeta, int N, int Ti, int Fi, void* stream);
/* ---- SyncBatchNorm halves (C2: `nn.SyncBatchNorm.convert_sync_batchnorm`, extras/decar-v2/main.py:82).  Every train-mode
 * BatchNorm of the path exists as "sums" + "finalize / apply" with the rank's sums in plain buffers in between, so the caller can
 * all-reduce them (RCCL) and pass the GLOBAL element count:
 *   forward : conv1_moments -> [all-reduce mom[0..53]] -> conv1_finalize;  conv3x3_fwd's sum / sumsq or colstats -> [all-reduce]
 *             -> bn_finalize(count = global)
 *   backward: bn_relu_pool_bwd_stats -> [all-reduce stat[0..127]] -> bn_relu_pool_bwd_apply;  conv1_bwd_sums -> [all-reduce lstat]
 *             -> conv1_bwd_finalize;  colbn_bwd_stats -> [all-reduce tmp] -> colbn_bwd_apply (+ add_d2f of the rank's own sums).
 * Parameter gradients always come from the rank's own sums (the data-parallel gradient all-reduce sums them afterwards). */
int audiossl_conv1_moments(const float* img, int N, int F, int T, double* mom, void* stream);
int audiossl_conv1_finalize(double* mom_totals, const float* w, const float* bias, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, double count, float* scale,
                            float* shift, float* save_mean, float* save_rstd, void* stream);
int audiossl_conv1_bwd_sums(int dtype, int conv_dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                            const float* gamma, const float* scale, const float* shift, const float* mean, const float* rstd,
                            const double* mom, const void* dP, const void* dxl, float* acc, float* lstat, void* stream);
int audiossl_conv1_bwd_finalize(const float* acc, const double* mom, const float* w, const float* bias, const float* gamma,
                                const float* mean, const float* rstd, double count_global, const float* gstat, float* dW,
                                float* dbias, float* dgamma, float* dbeta, void* stream);
int audiossl_bn_relu_pool_bwd_stats(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl,
                                    const float* scale, const float* shift, const float* mean, const float* rstd, float* stat,
                                    int N, int Ti, int Fi, void* stream);
int audiossl_bn_relu_pool_bwd_apply(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl,
                                    const float* scale, const float* shift, const float* mean, const float* rstd, float* stat,
                                    const float* gstat, double count_global, void* dY, float* dgamma, float* dbeta, int N, int Ti,
                                    int Fi, void* stream);
int audiossl_colbn_bwd_stats(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale,
                             const float* shift, const float* mean, const float* rstd, int relu, int groups, long M, int C,
                             double* tmp, void* stream);
int audiossl_colbn_bwd_apply(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale,
                             const float* shift, const float* mean, const float* rstd, int relu, int groups, long M, int C,
                             const double* sums, double count, void* da, void* stream);
int audiossl_im2col3x3(int dtype, const void* X, void* col, int N, int Ti, int Fi, void* stream);
int audiossl_pack_conv_w(int dtype, const float* W, void* Wf, void* Wd, void* stream);
/* the two 3x3 layers of one encoder (features_2.0 / features_3.0, audiontt.py:52-60) in one launch */
int audiossl_pack_conv_w2(int dtype, const float* Wa, void* Wfa, void* Wda, const float* Wb, void* Wfb, void* Wdb, void* stream);
int audiossl_unpack_conv_dw(const float* dWp, float* dW, void* stream);

/* Implicit-GEMM 3x3 / 64->64 convolution on the bf16 MFMA pipe (no im2col buffer); bf16 only, Fi in {32, 16}.
 * conv3x3_fwd : Y = conv(X, W) (+bias); W = packed [64][576] (pack_conv_w: Wf = forward, Wd = data gradient);
 *               optional BatchNorm batch statistics of the values as stored: sum / sumsq fp64 [64] (zeroed inside).
 *               stat_replicas > 1: sum -> [stat_replicas][128] doubles (replica r: sums at r*128, sums of squares at r*128 + 64;
 *               sumsq == sum + 64): workgroup i adds into replica i % stat_replicas, bn_relu_pool_train_fwd folds them.
 * conv3x3_wgrad: dWp fp32 [64][576] += dY^T * patches(X) (caller zeroes; unpack_conv_dw maps back to [co][ci][3][3]).
 *   workspace (optional, 256 * 64 * 576 floats = 37.7 MB covers every shape): per-workgroup results are stored there and
 *   folded by a second kernel; null = every workgroup adds its result to dWp with fp32 atomics. */
int audiossl_conv3x3_fwd(const void* X, const void* W, const float* bias, void* Y, int out_f32, double* sum,
                         double* sumsq, int stat_replicas, int N, int Ti, int Fi, void* stream);
int audiossl_conv3x3_wgrad(const void* dY, const void* X, float* dWp, float* workspace, long workspace_floats, int N, int Ti,
                           int Fi, void* stream);

/* ---- K9-K12 GEMM: every nn.Linear / matmul / einsum of the path ------------------------------------------
 * (audiontt.py:62-68; delores_s/upstream_expert.py:15-22, 36; delores_m/upstream_expert.py:250-252)
 * C[M,N] (+)= alpha * op(A) * op(B), operands of `dtype`, fp32 accumulate.
 *   trans_a = 0: A is [M][K] (lda)   trans_a = 1: A is [K][M] (lda)
 *   trans_b = 0: B is [N][K] (ldb)   trans_b = 1: B is [K][N] (ldb)     (torch Linear weights are [N][K])
 * Epilogue, in order: + bias[N]; ReLU; * keep[M][ldk] * keep_scale (dropout); zero where gate[M][ldg] <= 0;
 * + resid[M][ldr] (fp32, out-of-place residual connection of the transformer blocks; ksplit 1, not atomic);
 * store as dtype, or fp32 (out_f32), or accumulate into fp32 C: atomic = 1 by atomicAdd (required when ksplit > 1 or when
 * other launches may add to C concurrently), atomic = 2 by plain load-add-store (this launch owns C; ksplit 1).
 * ksplit > 1: the bias is added by the first split only; relu / keep / gate need the whole sum and are rejected. */
int audiossl_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K, float alpha, const void* A, long lda,
                  const void* B, long ldb, void* C, long ldc, const float* bias, int relu, const uint8_t* keep, long ldk,
                  float keep_scale, const void* gate, long ldg, int out_f32, int atomic, int ksplit, const float* resid,
                  long ldr, void* stream);

/* bf16 Linear -> ReLU -> Dropout (`src/encoder/audiontt.py:62-66`: fc block of AudioNTT2020Task6) with the keep mask DRAWN in the
 * epilogue: C (bf16) = dropout(relu(alpha op(A) op(B) + bias)) * keep_scale, element (row, col) kept iff the counter hash of
 * audiossl_dropout_mask says so for index row * N + col (seed' = (seed + *counter) mod 2^48; counter nullable) - the same mask, never
 * stored.  The backward gates on the stored output (gate argument of audiossl_gemm), it needs no mask either. */
int audiossl_gemm_dropout(int trans_a, int trans_b, int M, int N, int K, float alpha, const void* A, long lda, const void* B,
                          long ldb, void* C, long ldc, const float* bias, int relu, unsigned long long seed, float p,
                          const long long* counter, float keep_scale, void* stream);

/* `count` (<= 4) independent bf16 problems of one kind in one launch (the three Barlow heads of delores_m): same M,
 * transposes and epilogue (alpha, fp32 / accumulating output, split-K), per-problem N, K, operand pointers and leading
 * dimensions.  N, K, A, lda, B, ldb, C, ldc are HOST arrays of `count` entries. */
int audiossl_gemm_multi(int count, int trans_a, int trans_b, int M, const int* N, const int* K, float alpha,
                        const void* const* A, const long* lda, const void* const* B, const long* ldb, void* const* C,
                        const long* ldc, int out_f32, int atomic, int ksplit, void* stream);
/* gemm_multi whose fp32 results are weight gradients applied on the spot (no reference counterpart: the reference's optimizer.step()
 * reads .grad tensors, `delores_m/upstream_expert.py:306-313`): P[i] / Mom[i] ([M][Nv[i]] fp32, leading dimension ldp[i]) take
 * torch.optim.SGD's update in the epilogue - the arithmetic of audiossl_sgd_momentum bit for bit (common.h: sgd_step) - and
 * Shadow[i] (bf16, nullable) the copy of the new parameter; the gradient itself is never stored.  One K split. */
int audiossl_gemm_multi_sgd(int count, int trans_a, int trans_b, int M, const int* Nv, const int* K, float alpha,
                            const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* P,
                            float* const* Mom, void* const* Shadow, const long* ldp, float lr, float momentum, float weight_decay,
                            float grad_scale, const float* grad_scale_dev, void* stream);
/* The Barlow-twins cross-correlation of up to four heads with the loss folded into the epilogue (`delores_s/upstream_expert.py:118-131`,
 * `delores_m/upstream_expert.py:266-269`): c_h = alpha * A_h^T B_h, A_h / B_h = the two normalised views [K_h][D] (row-major, K_h =
 * batch rows); stores dc_h = dscale_h * (c_h - I) in bf16 [D][D] and adds coef_h * sum (c_h - I)^2 into loss_rep_h[0..31] (32 fp32
 * replicas the caller zeroes and sums: same-address atomics serialise).  c_h itself is never stored.  D % 64 == 0. */
int audiossl_gemm_multi_barlow(int count, int D, const int* K, float alpha, const void* const* A, const long* lda,
                               const void* const* B, const long* ldb, void* const* dc, const float* coef, const float* dscale,
                               float* const* loss_rep, void* stream);

/* ---- encoder tail: delores_s/upstream_encoder.py:26-28 ---------------------------------------------------- */
int audiossl_maxmean_fwd(int dtype, int out_f32, const void* H, void* y, uint8_t* arg, int N, int Tt, int D, void* stream);
int audiossl_maxmean_bwd(int dtype, int gdtype, const void* dy, const uint8_t* arg, const void* H, void* dA, int N, int Tt,
                         int D, void* stream);

/* ---- K10/K11 Barlow head: delores_s/upstream_expert.py:11-46, src/utils/utils.py:185-189 ------------------
 * colbn_fwd: h = act(scale_g*a+shift_g) on [G][M][C].  colbn_bwd: BatchNorm1d(train) backward per group (tmp = 2*G*C
 * doubles scratch), parameter grads summed over groups.
 * barlow_loss: loss += coef * sum (c - I)^2 ; dc = dscale * (c - I). */
int audiossl_colbn_fwd(int dtype, int adtype, const void* a, const float* scale, const float* shift, int relu, void* h,
                       int groups, long M, int C, void* stream);
/* colbn_train_fwd: colstats + bn_finalize + colbn_fwd of one train-mode BatchNorm1d (`nn.BatchNorm1d` inside the projector,
 * `src/upstream/delores_m/upstream_expert.py:18-29`) in a single launch for short batches (M <= 1024 rows per group,
 * C % 32 == 0): h = act(BN(a)), scale/shift/mean/rstd [G][C] saved for the backward, running statistics updated group
 * after group. */
int audiossl_colbn_train_fwd(int dtype, int adtype, const void* a, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, int relu, int groups,
                             long M, int C, void* h, float* scale, float* shift, float* save_mean, float* save_rstd,
                             void* stream);
int audiossl_colbn_bwd(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale, const float* shift, const float* mean,
                       const float* rstd, int relu, int groups, long M, int C, double* tmp, void* da, float* dgamma,
                       float* dbeta, void* stream);
/* Multi-problem forms (count <= 4 layers of one shape per launch, bf16 outputs) for the three Barlow heads p1-p3 of
 * DeLoRes-M (`src/upstream/delores_m/upstream_expert.py:133-135, 271`), which run the same chain on different operands.
 * The arrays are host arrays of device pointers (null array = null for every problem); stats[p] is [4][groups*C] fp32
 * (scale, shift, mean, rstd), written by the forward and read by the backward.  coef / dscale: host arrays. */
int audiossl_colbn_train_fwd_multi(int count, int adtype, const void* const* a, const float* const* gamma,
                                   const float* const* beta, float* const* running_mean, float* const* running_var,
                                   float momentum, float eps, int relu, int groups, long M, int C, void* const* h,
                                   float* const* stats, void* stream);
int audiossl_colbn_bwd_multi(int count, int adtype, int gdtype, const void* const* a, const void* const* dh,
                             const float* const* stats, int relu, int groups, long M, int C, void* const* da,
                             float* const* dgamma, float* const* dbeta, void* stream);
int audiossl_barlow_loss_multi(int count, const float* const* c, int D, const float* coef, const float* dscale,
                               void* const* dc, float* const* loss_out, void* stream);
int audiossl_add_d2f(const double* src, float* dst, int n, void* stream);
int audiossl_barlow_loss(int dtype, const float* c, int D, float coef, float dscale, void* dc, float* loss_out,
                         void* stream);

/* ---- K12/K13 MoCo head: delores_m/upstream_expert.py:147-172, 231-264, 270 -------------------------------- */
int audiossl_l2norm_fwd(int dtype, const float* q, int B, int D, void* qn, float* qn32, float* inv_norm, void* stream);
/* l2norm_fwd of q and of k and rowdot(qn, kn) * scale in one launch (the MoCo head's prologue) */
int audiossl_moco_prep(int dtype, const float* q, const float* k, int B, int D, float scale, void* qn, float* qn32, float* qinv,
                       void* kn, float* kn32, float* kinv, float* lpos, void* stream);
int audiossl_rowdot(const float* a, const float* b, int B, int D, float scale, float* out, void* stream);
int audiossl_moco_ce_fwd(const float* lpos, const float* lneg, int B, int K, float* lse, float* loss_out, void* stream);
int audiossl_moco_ce_bwd(int dtype, const float* lpos, const float* lneg, const float* lse, int B, int K, float gscale,
                         void* P, float* dlpos, void* stream);
/* The same head without the [B][K] logits in memory (bf16 path): the logits GEMM qn [B][dim] x queue [dim][K] / T reduces
 * its own tile - mode 1: part [B][ceil(K/64)][2] fp32 = (max, sum exp) per 64-column slab; moco_lse_merge folds them with
 * lpos into lse / loss / dlpos; mode 2: P [B][K] bf16 = exp(logit - lse) * gscale, recomputed for the dq GEMM.
 * Replaces `logits = cat([l_pos, l_neg]) / T; F.cross_entropy(logits, 0)` (delores_m/upstream_expert.py:250-264, 270). */
int audiossl_moco_logits(int mode, const void* qn, const void* queue, int B, int K, int dim, float inv_t, float* part,
                         const float* lse, float gscale, void* P, void* stream);
int audiossl_moco_lse_merge(const float* lpos, const float* part, int B, int nslot, float gscale, float* lse,
                            float* loss_out, float* dlpos, void* stream);
int audiossl_l2norm_bwd(int dtype, const float* dqn, const float* dlpos, const float* kn32, const float* qn32,
                        const float* inv_norm, int B, int D, void* dq, void* stream);
/* ptr_dev (optional, int64 on the device = the reference's `queue_ptr` buffer): when given, the write position is read
 * from it and advanced by B (mod K) on the device, so the call can sit inside a captured hipGraph; `ptr` is then ignored. */
int audiossl_enqueue(int dtype, const float* keys, int B, int D, int K, int ptr, long long* ptr_dev, float* queue, void* shadow,
                     void* stream);
/* shadow (optional): bf16 [n], the updated pk as the key encoder's MFMA operands, written in the same pass */
int audiossl_ema_update(float* pk, const float* pq, long n, float m, void* shadow, void* stream);

/* Host-side switch: 1 = every scratch pointer handed to the entry points below is already zero (the caller cleared its whole
 * scratch arena with one memset), so they skip their own hipMemsetAsync; 0 (default) = they zero their scratch themselves. */
int audiossl_set_prezeroed(int on);
/* Host-side query (diagnostics): the kernel the most recent audiossl_gemm / gemm_multi / gemm_multi_barlow / moco_logits call
 * launched, as a substring of the name rocprofv3 prints for it ("gemm_p6_multi_kernel<false, false>"; instantiations on the bf16
 * type appear mangled there: "gemm_multi_kernelIDF16bLb0ELb0ELi64ELi2ELi4E").  The dispatch picks among several instantiations
 * by shape; bench.py's per-kernel roofline table uses this to join its rows to profiles/ kernel_stats.csv. */
int audiossl_last_kernel(char* name, int capacity);

/* ---- K17 optimiser + plumbing: delores_s/upstream_expert.py:236-243 (torch.optim.SGD) ---------------------- */
/* sgd_momentum: optional tail work of the same pass - shadow_bf16 (nullable): bf16 copy of the updated parameters (what
 * `cast` would produce at the start of the next step); zero_grad: clear g after reading it. */
int audiossl_sgd_momentum(float* p, float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                          int first, float grad_scale, const float* grad_scale_dev, void* shadow_bf16, int zero_grad,
                          void* stream);
/* sgd_momentum over nseg segments of the flat buffers (segs: DEVICE table of (offset, length) pairs in elements, multiples of 4;
 * max_n = the longest; shadow_bf16 nullable, indexed like p): the remaining tensors of a slice whose large ones gemm_multi_sgd updated. */
int audiossl_sgd_momentum_segments(float* p, float* g, float* buf, const long* segs, int nseg, long max_n, float lr, float momentum,
                                   float weight_decay, int first, float grad_scale, const float* grad_scale_dev, void* shadow_bf16,
                                   int zero_grad, void* stream);
/* g[segs[2s] .. + segs[2s+1]) = 0 for nseg segments (segs: DEVICE table of (offset, length) pairs in elements, max_n = the longest):
 * clears the small tensors of a flat gradient whose large ones are stored, not accumulated, by their single writer
 * (src/optim.py: HipSGD.step_tail(keep_stale=...)); no reference counterpart - the reference's optimizer.zero_grad() clears all. */
int audiossl_zero_segments(float* g, const long* segs, int nseg, long max_n, void* stream);
int audiossl_cast(int dtype, const float* src, void* dst, long n, void* stream);
int audiossl_cast_back(int dtype, const void* src, float* dst, long n, void* stream);
/* keep[i] = splitmix64(seed', i) >= p; seed' = (seed + *counter) mod 2^48 when `counter` (device int64) is given, so that
 * a replayed hipGraph draws a fresh mask every step. */
int audiossl_dropout_mask(uint8_t* keep, long n, unsigned long long seed, float p, const long long* counter, void* stream);
/* out = g where h > 0 else 0 (nn.ReLU backward on the stored activation); eval-mode BatchNorm folded to scale/shift
 * (audiontt.py:47,53,58 under model.eval()); y += a*x on fp32 buffers. */
int audiossl_relu_bwd(int dtype, const void* g, const void* h, void* out, long n, void* stream);
int audiossl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                            float eps, int C, float* scale, float* shift, void* stream);
int audiossl_axpy(float* y, const float* x, float a, long n, void* stream);
/* x = hi + lo, hi = bf16(x), lo = bf16(x - hi): the projector's first GEMM runs on both pieces, because the time-pooled
 * features it reads have |mean| >> batch-std and a single bf16 rounding would eat their batch variation. */
int audiossl_split_bf16(const float* x, void* hi, void* lo, long n, void* stream);
/* Default bf16 path, same problem at no extra GEMM: yc = bf16(y - column mean per group) [G][M][C], cmean [G][C] fp32
 * (M <= 1024, C % 32 == 0).  The projector's first Linear has no bias and is followed by train-mode BatchNorm
 * (`src/upstream/delores_s/upstream_expert.py:15-22`), which removes the per-column constant the shift produces; the weight
 * gradient is unchanged because a BatchNorm input gradient sums to zero over the batch.  shift_running_mean repairs the
 * one thing that does see the shift: running_mean[j] += sum_g m (1-m)^(G-1-g) cmean[g] . W[j]   (W bf16 [D][K]). */
int audiossl_center_cast(const float* y, void* yc, float* cmean, int groups, long M, int C, void* stream);
int audiossl_shift_running_mean(const void* W, const float* cmean, float* running_mean, int D, int K, int groups,
                                float momentum, void* stream);
/* the same for up to four heads per launch (host arrays of device pointers / widths, as the other *_multi entry points) */
int audiossl_center_cast_multi(int count, const float* const* y, void* const* yc, float* const* cmean, const int* C, int groups,
                               long M, void* stream);
int audiossl_shift_running_mean_multi(int count, const void* const* W, const float* const* cmean, float* const* running_mean,
                                      int D, const int* K, int groups, float momentum, void* stream);

/* ---- K14 NT-Xent / ClusterLoss: extras/slicer/contrastive_loss.py:6-92 -----------------------------------------
 * sim [N][N] fp32 = z z^T / tau from audiossl_gemm (N = 2B; positives at (r + B) mod N; the diagonal is excluded).
 * fwd: lse[r] and loss += mean_r(lse_r - sim[r][pos]).  bwd: dsim = (softmax - onehot_pos) * gscale, 0 on the diagonal. */
int audiossl_ntxent_fwd(const float* sim, int N, int B, float* lse, float* loss_out, void* stream);
int audiossl_ntxent_bwd(int dtype, const float* sim, const float* lse, int N, int B, float gscale, void* dsim, void* stream);

/* ---- K15/K16 DeepCluster-v2: extras/decar-v2/utils.py:291-318 (k-means E/M steps), main.py:228-233 (prototype CE) ----
 * row_argmax: assign[r] = first argmax of dot[r][:] (dot = mem * centroids^T from audiossl_gemm).
 * kmeans_accumulate: sums[a] += x[r], counts[a] += 1 (zeroed inside).  kmeans_update: mean + L2-normalise, empty
 * clusters keep their previous centroid.  ce_rows: CrossEntropyLoss(ignore_index) forward + dlogits. */
int audiossl_row_argmax(const float* dot, long N, int K, long long* assign, void* stream);
int audiossl_kmeans_accumulate(const float* x, const long long* assign, long N, int K, int D, float* sums, int* counts,
                               void* stream);
int audiossl_kmeans_update(const float* sums, const int* counts, int K, int D, float* centroids, void* stream);
/* Offline pseudo-labeler (extras/decar-v2/clustering.py:19-115: faiss PCA-whitening + k-means): the Lloyd M step with plain
 * means (empty clusters keep their centroid) and out[r] = scale * |x_r|^2, the bias that turns the dot-product GEMM + row_argmax
 * into the Euclidean E step. */
int audiossl_kmeans_update_mean(const float* sums, const int* counts, int K, int D, float* centroids, void* stream);
int audiossl_row_sqnorm(const float* x, int rows, int D, float scale, float* out, void* stream);
int audiossl_ce_rows(int dtype, const float* logits, const long long* target, int B, int K, int ignore_index, int* cnt,
                     float* loss_out, void* dlogits, void* stream);

/* ---- row softmax: nn.Softmax(dim=1) closing SLICER's cluster projector (src/upstream/slicer/upstream_encoder.py:15-20);
 * fp32 [M][C]; bwd: gx = y * (gy - sum_c gy*y). */
int audiossl_softmax_rows_fwd(const float* x, float* y, int M, int C, void* stream);
int audiossl_softmax_rows_bwd(const float* y, const float* gy, float* gx, int M, int C, void* stream);

/* ---- K18 AST / MAST transformer encoder (BASELINE config 4, "AST-base 12x768"): the ViT timm builds for ASTModel,
 * extras/mast_new/mast/models/ast_work.py:70-81, 101, 183-230; optimiser extras/mast_new/mast/moco_model.py:373-379 ----
 * attn_fwd: qkv bf16 [B*S][3*H*64] (q | k | v column blocks, head h at columns h*64) -> out bf16 [B*S][H*64] =
 *           softmax(scale * q k^T) v per (clip, head), lse fp32 [B*H][S] kept for the backward.  S <= 128 runs as one
 *           tile per (clip, head); longer sequences (10 s clips: 1,212 tokens) walk 128-token blocks, B*H <= 65535.
 * attn_bwd: dout bf16 [B*S][H*64] -> dqkv bf16 [B*S][3*H*64].  `out` = what attn_fwd wrote (needed when S > 128: the
 *           multi-block backward takes D = rowsum(dout * out) from it; may be NULL for S <= 128).
 * layernorm_fwd: x fp32 [M][C] -> y bf16 (+ the same in fp32 into y32 when it is not NULL), mean / rstd fp32 [M] (C <= 1024;
 *            the MViT widths are 96 * 2^k).
 * layernorm_bwd: dres fp32 [M][C] += dx (the residual-stream gradient accumulates in place); dgamma, dbeta += .
 * gelu_fwd / gelu_bwd: exact GELU on bf16, da = dh * gelu'(a).
 * patch_unfold: x fp32 [B][F][T] -> bf16 [B*nf*nt][256], 16x16 patches, strides (fstride, tstride), row = (b, pf, pt).
 * tile_rows: out[r] = scale * src[(r / group) % period]: position embedding tiled over the batch (group 1, period N), and
 *            the backward of the mean over tokens (group N, period B, scale 1/N).
 * adamw: torch.optim.AdamW on a flat buffer; `step` = device int64 holding the 1-based step count. */
int audiossl_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, float scale, void* stream);
int audiossl_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S, int H,
                      float scale, void* stream);
int audiossl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* y32, float* mean, float* rstd,
                           int M, int C, float eps, void* stream);
int audiossl_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           float* dres, float* dgamma, float* dbeta, int M, int C, void* stream);
int audiossl_gelu_fwd(const void* a, void* h, long n, void* stream);
int audiossl_gelu_bwd(const void* a, const void* dh, void* da, long n, void* stream);
int audiossl_patch_unfold(const float* x, void* out, int B, int F, int T, int fstride, int tstride, void* stream);
int audiossl_tile_rows(const float* src, float* out, long rows, int period, int group, float scale, int C, void* stream);
int audiossl_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, const long long* step, void* stream);

/* ---- MViTv2 pooling attention (the encoder SS-MAST instantiates: models_msn.py:147 -> ASTModel(model_size='mvit')):
 * extras/mast_new/mast/mvit/models/attention.py:12-41 (attention_pool), :44-90 (cal_rel_pos_spatial), :93-302
 * (MultiScaleAttention), :304-393 (MultiScaleBlock).  GEMMs are audiossl_gemm, LayerNorm over the width audiossl_layernorm_*.
 * mvit_pool_fwd: one of q / k / v out of qkv bf16 [B*H*W][ldq] (column block starting at col0, head h at col0 + h*d) ->
 *           out fp32 [B][heads][Ho*Wo][d] = LayerNorm_d(depthwise 3x3 conv, stride (sh, sw), pad 1, filters w [d][3][3] shared by
 *           the heads); z / mean / rstd keep the conv output and its row statistics for the backward.  w == NULL: no pooling
 *           on this path - the head split only (Ho = H, Wo = W, no LayerNorm).  d <= 128.
 * mvit_pool_bwd: dout fp32 [B][heads][Ho*Wo][d] -> the tensor's column block of dqkv bf16 [B*H*W][ldq] (written, not
 *           accumulated); dw [d][3][3], dgamma, dbeta accumulated; dz = scratch of dout's shape.
 * mvit_attn_fwd: q [B][heads][qh*qw][d], k / v [B][heads][kh*kw][d] fp32 -> out bf16 [B*qh*qw][heads*d] =
 *           softmax(scale q k^T + q . rh[ih[query row][key row]] + q . rw[iw[query col][key col]]) v (+ q when residual: the
 *           residual pooling connection), lse fp32 [B][heads][qh*qw].  rh / rw fp32 [nrh | nrw][d] (both or neither), ih / iw
 *           int32 [qh][kh], [qw][kw] row tables (device).  d in {64, 96, 128}.
 * mvit_attn_bwd: dout bf16 [B*Lq][heads*d] -> dq fp32 (written), dk / dv fp32 and drh / drw (accumulated: zero them first).
 * tokpool_max_fwd / _bwd: MaxPool2d((kh, kw), (sh, sw), padding (kh/2, kw/2)) over the token grid of x fp32 [B][H*W][C]
 *           (the skip path of a block whose queries are pooled, attention.py:343-350); arg uint8 = tap of the maximum;
 *           bwd writes dx (gathered per input token). */
int audiossl_mvit_pool_fwd(const void* qkv, int ldq, int col0, const float* w, const float* gamma, const float* beta, float* out,
                           float* z, float* mean, float* rstd, int B, int heads, int d, int H, int W, int Ho, int Wo, int sh, int sw,
                           float eps, void* stream);
int audiossl_mvit_pool_bwd(const void* qkv, int ldq, int col0, const float* w, const float* gamma, const float* dout, const float* z,
                           const float* mean, const float* rstd, float* dz, float* dw, float* dgamma, float* dbeta, void* dqkv, int B,
                           int heads, int d, int H, int W, int Ho, int Wo, int sh, int sw, void* stream);
int audiossl_mvit_attn_fwd(const float* q, const float* k, const float* v, const float* rh, const float* rw, const int* ih,
                           const int* iw, void* out, float* lse, int B, int heads, int d, int qh, int qw, int kh, int kw, int nrh,
                           int nrw, int residual, float scale, void* stream);
int audiossl_mvit_attn_bwd(const float* q, const float* k, const float* v, const float* rh, const float* rw, const int* ih,
                           const int* iw, const void* dout, const float* lse, float* dq, float* dk, float* dv, float* drh, float* drw,
                           int B, int heads, int d, int qh, int qw, int kh, int kw, int nrh, int nrw, int residual, float scale,
                           void* stream);
int audiossl_tokpool_max_fwd(const float* x, float* y, void* arg, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                             void* stream);
int audiossl_tokpool_max_bwd(const float* dy, const void* arg, float* dx, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                             void* stream);

/* ---- LARS: extras/delores-s/multi_proc.py:4-43, on the flat parameter buffer ---------------------------------------
 * seg: n_seg x {int64 offset, int64 numel, int32 flags (bit0 weight decay, bit1 trust ratio), int32 pad} (device);
 * lr: per-segment learning rate (device); norms: 2*n_seg doubles of scratch. */
int audiossl_lars_step(float* p, const float* g, float* mu, const void* seg, int n_seg, const float* lr, float weight_decay,
                       float momentum, float eta, float grad_scale, double* norms, void* stream);
/* apex LARC around torch.optim.SGD (`extras/decar-v2/main.py:92-97, 111`: SGD(momentum 0.9, wd 1e-6) wrapped in
 * LARC(trust_coefficient 0.001, clip False)): per tensor alr = tc |p| / (|g| + wd |p| + eps) when both norms are non-zero,
 * g' = (g + wd p) alr, else g' = g; then buf = m buf + g', p -= lr buf.  seg as lars_step; flags & 4 = no gradient this step
 * (the reference sets p.grad = None for frozen prototypes: the tensor and its momentum are left untouched). */
int audiossl_larc_step(float* p, const float* g, float* mu, const void* seg, int n_seg, float lr, float weight_decay,
                       float momentum, float trust_coefficient, float eps, int clip, float grad_scale, double* norms,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOSSL_HIP_H */
