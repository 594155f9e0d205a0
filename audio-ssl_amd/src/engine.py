"""Launch sequences of the upstream step on one GPU (host orchestration of libaudiossl_hip.so).

Explicit forward / backward of
  * AudioNTT2020Task6            (`src/encoder/audiontt.py:72-104`)
  * max_T + mean_T pooling       (`src/upstream/delores_s/upstream_encoder.py:26-28`)
  * the Barlow projector + loss  (`src/upstream/delores_s/upstream_expert.py:11-46`)
  * the MoCo InfoNCE head        (`src/upstream/delores_m/upstream_expert.py:231-264, 270`)
as sequences of C-ABI calls.  torch is used for device buffers and streams only; no ATen arithmetic sits on
the path.  Gradients are ACCUMULATED into the tensors handed in (flat-buffer views in training).

Activation layout [N][T][F][64] (time, mel, channel); `dtype` 0 = fp32 storage / exact-f32 MFMA,
1 = bf16 storage / bf16 MFMA with fp32 accumulate (parameters, statistics, losses stay fp32).
"""
import os

import torch

from src import _native as N

import os as _os
_STEM_VALU = _os.environ.get("AUDIOSSL_STEM_VALU") == "1"     # debug switch: fp32 VALU stem on the bf16 path
BN_MOMENTUM = 0.1
BN_EPS = 1e-5
_BN_FUSED = _os.environ.get("AUDIOSSL_BN_FUSED", "1") != "0"


# AUDIOSSL_ONE_STREAM=1: no side streams at all - every launch of a step goes to the current stream in program order
ONE_STREAM = os.environ.get("AUDIOSSL_ONE_STREAM", "0") == "1"


# ---- stream forks under hipGraph capture ------------------------------------------------------------------------------------
# Capture rules (HIP = CUDA): a stream joins a capture by waiting on an event recorded in a capturing stream (a FORK); every
# stream that joined must be ordered back into the capture's ORIGIN stream before hipStreamEndCapture (a JOIN); no wait on an
# event recorded outside the capture.  Every fork of this package's steps starts at the origin stream (key encoder, loss heads,
# preparation sweeps, weight gradients) and is joined back into it.  The one variant of round 2 that forked from an already forked
# stream (the heads' weight-gradient launches sent to the WGRAD stream from inside the heads' stream, joined into the origin by
# the encoder backward's WGRAD join - legal by the rules above, and fine when issued eagerly) died inside hipStreamEndCapture on
# ROCm 7.2 with a segmentation fault instead of an error code.  Nothing in these sources can tell a runtime defect from a rule this
# runtime enforces differently, so the product simply never builds that topology: while a capture scope is open, `fork` refuses a
# parent stream that is not the scope's origin (AUDIOSSL_ALLOW_NESTED_FORK=1 lifts the check for experiments).
_CAPTURE_ORIGIN = []


class capture_scope:
    """with capture_scope(): ... - entered INSIDE `torch.cuda.graph(...)` by the graph steps; the stream current at entry is the
    capture's origin."""

    def __enter__(self):
        _CAPTURE_ORIGIN.append(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        _CAPTURE_ORIGIN.pop()
        return False


def fork(side, parent=None, device=None):
    """Order `side` after everything enqueued so far on `parent` (default: the current stream) - the only way this package
    starts work on a side stream."""
    parent = parent if parent is not None else torch.cuda.current_stream(device)
    if _CAPTURE_ORIGIN and parent.cuda_stream != _CAPTURE_ORIGIN[-1].cuda_stream and side.cuda_stream != _CAPTURE_ORIGIN[-1].cuda_stream \
            and os.environ.get("AUDIOSSL_ALLOW_NESTED_FORK", "0") != "1":
        raise RuntimeError("nested stream fork inside a hipGraph capture: a side stream may only be forked from the capture's origin "
                           "stream (a fork from an already forked stream crashed hipStreamEndCapture on ROCm 7.2 - engine.py, "
                           "DESIGN.md section 7); restructure the step or set AUDIOSSL_ALLOW_NESTED_FORK=1 to experiment")
    side.wait_stream(parent)


class SideStream:
    """A second HIP stream for work that is off the critical dependency chain (weight-gradient GEMMs, the key
    encoder): `run(fn)` orders the side stream after everything enqueued so far on the current stream, `join()`
    orders the current stream after the side stream."""

    def __init__(self):
        self._streams = {}

    def stream(self, device):
        if ONE_STREAM:
            return torch.cuda.current_stream(device)
        key = str(device)
        if key not in self._streams:
            # (HIP stream priorities were tried for the dependency-chain streams - capture stream, key encoder, loss heads at -1:
            # the B = 512 step went from 2.05 to 2.58 ms, so every stream keeps the default priority)
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def run(self, device, fn):
        if ONE_STREAM:
            return fn()
        s = self.stream(device)
        fork(s, torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            return fn()

    def join(self, device):
        if not ONE_STREAM:
            torch.cuda.current_stream(device).wait_stream(self.stream(device))


WGRAD = SideStream()


def _empty(shape, dtype, like=None, device=None):
    return torch.empty(shape, dtype=dtype, device=device if device is not None else like.device)


class ZeroArena:
    """Zero-initialised scratch for one fused training step.  The entry points that accumulate into caller-provided scratch
    (column / tap statistics, split-K targets, atomically reduced partials) each used to clear it with their own memset:
    ~30 tiny nodes per step, which cost more on the launch side of a graph replay (~15 us of host time per node) than on
    the GPU.  Inside `with ARENA.step(device):` every such buffer is a slice of one persistent buffer cleared by ONE memset,
    and the library is told to skip its own (`audiossl_set_prezeroed`)."""

    def __init__(self, nbytes=8 << 20):
        self.nbytes, self.buf, self.off, self.active = nbytes, None, 0, False

    def step(self, device):
        arena = self

        class _Ctx:
            def __enter__(self_):
                if arena.buf is None or arena.buf.device != device:
                    arena.buf = torch.empty(arena.nbytes, dtype=torch.uint8, device=device)
                arena.off, arena.active = 0, True
                arena.buf.zero_()
                N.call_host("set_prezeroed", 1)

            def __exit__(self_, *exc):
                arena.active = False
                N.call_host("set_prezeroed", 0)
                return False
        return _Ctx()

    def zeros(self, shape, dtype, like=None, device=None):
        """A zeroed tensor: an arena slice inside a step, `torch.zeros` otherwise (or when the arena is exhausted)."""
        dev = device if device is not None else like.device
        if self.active and self.buf.device == dev:
            n = 1
            for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
                n *= int(d)
            nb = n * torch.empty((), dtype=dtype).element_size()
            if self.off + nb <= self.nbytes:
                t = self.buf[self.off:self.off + nb].view(dtype).view(shape)
                self.off += (nb + 255) // 256 * 256
                return t
        return torch.zeros(shape, dtype=dtype, device=dev)

    def scratch(self, shape, dtype, like):
        """Scratch that the callee clears itself unless it has been told the caller did: zeroed inside a step, raw outside."""
        return self.zeros(shape, dtype, like=like) if self.active else _empty(shape, dtype, like=like)


ARENA = ZeroArena()


def fast_copy(dst, src):
    """dst <- src (fp32, contiguous, same shape) by an elementwise kernel.  `Tensor.copy_` turns a device-to-device copy into the
    runtime's blit kernel (`__amd_rocclr_copyBuffer`), which took 72 us for the 13 MB of a view batch at the head of the main
    stream of every step (rocprofv3 timeline, round 3) - 0.4 TB/s; the float4 kernel moves it at HBM speed."""
    n = src.numel()
    if src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and n % 1024 == 0 \
            and n // 1024 < (1 << 31) and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0:
        N.call("tile_rows", src, dst, n // 1024, n // 1024, 1, 1.0, 1024)
    else:
        dst.copy_(src)
    return dst


def _ksplit(M, Nn, K, target=512):
    """Split-K factor of an accumulating (fp32 atomics) GEMM.  Splitting pays only while every workgroup keeps a long K loop
    (>= 32 steps of 64) and the grid stays within two workgroups per CU: 2048x2048x6144 78 us at 2 vs 93 us unsplit, but
    2048x2048x1024 28 us unsplit vs 37 / 62 us at 2 / 4 (tools/gemm_floor.py)."""
    tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
    return int(max(1, min(target // max(tiles, 1), (K // 64) // 32)))


def gemm(dtype, ta, tb, M, Nn, K, A, lda, B, ldb, C, ldc, alpha=1.0, bias=None, relu=0, keep=None, ldk=0, keep_scale=1.0,
         gate=None, ldg=0, out_f32=0, atomic=0, ksplit=1, resid=None, ldr=0):
    N.call("gemm", dtype, ta, tb, M, Nn, K, float(alpha), A, lda, B, ldb, C, ldc, bias, relu, keep, ldk, float(keep_scale),
           gate, ldg, out_f32, atomic, ksplit, resid, ldr)


def linear_fwd(dtype, X, W, M, Nout, K, bias=None, relu=0, keep=None, keep_scale=1.0, out=None, out_f32=0):
    """X [M,K] @ W[Nout,K]^T (+bias) -> [M,Nout]"""
    if out is None and dtype == N.BF16 and out_f32 and not relu and keep is None and M * Nout <= (1 << 16) and K >= 1024 and K % 512 == 0:
        # tiny outputs with a long contraction (the 2048 -> 128 embedding layers of the MoCo heads): eight workgroups walking 32
        # K-steps each took 17-22 us of pure latency on the critical path; K is split over the idle CUs instead - fp32 atomics
        # into a zeroed output, the bias added by the first split
        out = ARENA.zeros((M, Nout), torch.float32, like=X)
        gemm(dtype, 0, 0, M, Nout, K, X, K, W, K, out, Nout, bias=bias, out_f32=1, atomic=1, ksplit=min(8, K // 256))
        return out
    if out is None:
        out = _empty((M, Nout), torch.float32 if (out_f32 or dtype == 0) else torch.bfloat16, like=X)
    gemm(dtype, 0, 0, M, Nout, K, X, K, W, K, out, Nout, bias=bias, relu=relu, keep=keep, ldk=Nout, keep_scale=keep_scale,
         out_f32=out_f32)
    return out


def linear_bwd_w(dtype, dY, X, dW, M, Nout, K, dw_zero=False):
    """dW[Nout,K] += dY[M,Nout]^T @ X[M,K]   (split-K, fp32 atomics).  dw_zero: the caller guarantees dW is zero and has this one
    writer in the step - an unsplit bf16 launch then stores its result instead of reading the zeros to add to them."""
    # every dW of the step is written by one stream at a time, so without split-K the accumulation needs no atomics
    ks = _ksplit(Nout, K, M)
    atomic = 1 if (ks > 1 or dtype != N.BF16) else (0 if dw_zero else 2)
    gemm(dtype, 1, 1, Nout, K, M, dY, Nout, X, K, dW, K, out_f32=1, atomic=atomic, ksplit=ks)


def linear_bwd_x(dtype, dY, W, M, Nout, K, alpha=1.0, gate=None, out=None, out_f32=0):
    """dX[M,K] = alpha * dY[M,Nout] @ W[Nout,K]   (optionally zeroed where gate <= 0).
    out_f32: store the result in fp32 - used whenever dX feeds a BatchNorm / pooling backward (precision contract in
    DESIGN.md section 6: such gradients are GEMM outputs, so fp32 is free; pure MFMA operands stay in `dtype`)."""
    if out is None:
        out = _empty((M, K), torch.float32 if (out_f32 or dtype == N.F32) else torch.bfloat16, like=dY)
    gemm(dtype, 0, 1, M, K, Nout, dY, Nout, W, K, out, K, alpha=alpha, gate=gate, ldg=K, out_f32=out_f32)
    return out


GD = N.F32     # storage type of gradients that enter a BatchNorm / pooling backward (free: they are GEMM outputs)

# "bf16_hp" (run.precision): additionally keep BatchNorm INPUTS (conv outputs, projector pre-activations) and the
# time-pooled features in fp32 and run the projector's first GEMM on hi+lo bf16 pieces.  Costs ~0.6 ms per 512-clip step
# (fp32 conv outputs are 2x the bytes), buys 2-3x smaller gradient deviation from the fp32 path.
HP = False


def set_high_precision(flag):
    global HP
    HP = bool(flag)


# SyncBatchNorm (`nn.SyncBatchNorm.convert_sync_batchnorm`, extras/decar-v2/main.py:82): when set, every train-mode BatchNorm of
# the encoder / BatchNorm1dFn exchanges its sums between the ranks (one small all-reduce per layer and direction) and normalises
# with the statistics of the GLOBAL batch.  Eager steps only (a collective cannot sit inside a captured graph).
SYNC_BN = None


class SyncBN:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)

    def all_reduce(self, t):
        self.dist.all_reduce(t, group=self.group)
        return t


def set_sync_bn(sync):
    global SYNC_BN
    SYNC_BN = sync if (sync is not None and sync.world > 1) else None


def _ad(dtype):
    """storage type of BatchNorm inputs"""
    return N.F32 if (HP or dtype == N.F32) else dtype


def pooled_dtype(dtype):
    """torch dtype of the time-pooled features handed to the projector: fp32 on every path.  The bf16 paths turn them into
    MFMA operands themselves (centred cast by default, hi + lo split under bf16_hp) - see projector_forward."""
    return torch.float32


def center_cast(Y, groups, B):
    """Y fp32 [groups*B, C] -> (bf16(Y - per-group column mean), column means [groups, C])"""
    C = Y.shape[1]
    yc = _empty((groups * B, C), torch.bfloat16, like=Y)
    cmean = _empty((groups, C), torch.float32, like=Y)
    N.call("center_cast", Y, yc, cmean, groups, B, C)
    return yc, cmean


def colsum_add(dtype, X, M, C, dst, tmp=None):
    """dst[C] += column sums of X [M,C]"""
    tmp = tmp if tmp is not None else ARENA.scratch((C,), torch.float64, X)
    N.call("colstats", dtype, X, 1, M, C, C, 0, tmp, None)
    N.call("add_d2f", tmp, dst, C)


def cast(dtype, src, n=None):
    """fp32 parameter -> tensor of the activation dtype (identity for fp32)."""
    if dtype == N.F32:
        return src
    out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    N.call("cast", dtype, src, out, src.numel())
    return out


# =============================================================================================== encoder
class EncoderCtx:
    """Everything the backward needs from one forward call."""
    pass


def _bn_train(dtype, Y, M, C, gamma, beta, rm, rv, update_running, groups=1):
    """Batch statistics of Y [groups][M][C] -> per-group (scale, shift, mean, rstd), each [groups*C]."""
    sq = ARENA.scratch((2, groups * C), torch.float64, Y)
    N.call("colstats", dtype, Y, groups, M, C, C, 1, sq[0], sq[1])
    if SYNC_BN is not None:
        SYNC_BN.all_reduce(sq)
        M = M * SYNC_BN.world
    st = _empty((4, groups * C), torch.float32, like=Y)
    N.call("bn_finalize", sq[0], sq[1], groups, float(M), C, gamma, beta, rm if update_running else None,
           rv if update_running else None, BN_MOMENTUM, BN_EPS, st[0], st[1], st[2], st[3])
    return st[0], st[1], st[2], st[3]


def bn_train_apply(dtype, ad, a, M, C, gamma, beta, rm, rv, update_running, groups, relu, out):
    """Train-mode BatchNorm1d of a [groups][M][C] (+ReLU) -> out; returns (scale, shift, mean, rstd).  One fused launch for
    the short projector batches, colstats -> bn_finalize -> colbn_fwd otherwise."""
    if M <= 1024 and C % 32 == 0 and _BN_FUSED and SYNC_BN is None:
        st = _empty((4, groups * C), torch.float32, like=a)
        N.call("colbn_train_fwd", dtype, ad, a, gamma, beta, rm if update_running else None, rv if update_running else None,
               BN_MOMENTUM, BN_EPS, relu, groups, M, C, out, st[0], st[1], st[2], st[3])
        return st[0], st[1], st[2], st[3]
    st = _bn_train(ad, a, M, C, gamma, beta, rm, rv, update_running, groups)
    N.call("colbn_fwd", dtype, ad, a, st[0], st[1], relu, out, groups, M, C)
    return st


def _bn_eval(like, C, gamma, beta, rm, rv):
    scale, shift = _empty((C,), torch.float32, like=like), _empty((C,), torch.float32, like=like)
    N.call("bn_eval_affine", gamma, beta, rm, rv, BN_EPS, C, scale, shift)
    return scale, shift


_WGRAD_WS = {}


def _wgrad_workspace(device):
    """Per-device scratch for the two-stage conv weight gradient (per-workgroup results + fold): 37.7 MB, allocated once
    (before any graph capture: the eager priming steps come first).  Launches that share it are ordered on one stream."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(256 * 64 * 576, dtype=torch.float32, device=device)
    return ws


STAT_REPLICAS = 8
LOSS_REPLICAS = 32          # replicas of a Barlow loss term accumulated by the correlation GEMM's epilogue (gemm_multi_barlow)
_BN_STATS_P = os.environ.get("AUDIOSSL_BN_STATS_P", "1") != "0"  # 0: BatchNorm-backward sums from a first sweep over the conv output
_DP1_BF16 = os.environ.get("AUDIOSSL_DP1_BF16", "1") != "0"      # 0: the gradient into the stem backward stays fp32


def _conv_block_fwd(dtype, Pin, Nimg, Ti, Fi, W, bias, bn, train, update_running, col, packed=None):
    """3x3 conv -> BN -> ReLU -> pool.  bf16: implicit-GEMM kernel with the batch statistics fused into its epilogue;
    fp32 (validation path): im2col + exact-f32 MFMA GEMM + colstats.  Returns (Y, Pout, stats, Wf, Wd)."""
    td = N.torch_dtype(dtype)
    if packed is not None:
        Wf, Wd = packed
    else:
        Wf, Wd = _empty((64, 576), td, like=Pin), _empty((64, 576), td, like=Pin)
        N.call("pack_conv_w", dtype, W, Wf, Wd)
    M = Nimg * Ti * Fi
    ad = _ad(dtype)
    Y = _empty((M, 64), N.torch_dtype(ad), like=Pin)          # BatchNorm input: fp32 on the fp32 and bf16_hp paths
    gamma, beta, rm, rv = bn
    fused = dtype == N.BF16 and Fi in (16, 32)
    # batch statistics: 8 replicas of the (sum | sum of squares) accumulator when the pooling launch folds them itself
    # (256 workgroups adding into one set of 128 addresses cost ~12 us at the end of the convolution)
    rep = STAT_REPLICAS if (fused and train and SYNC_BN is None) else 1
    sq = ARENA.scratch((rep, 2, 64), torch.float64, Pin) if (fused and train) else None
    if fused:
        N.call("conv3x3_fwd", Pin, Wf, bias, Y, int(ad == N.F32), None if sq is None else sq[0, 0], None if sq is None else sq[0, 1],
               rep, Nimg, Ti, Fi)
    else:
        N.call("im2col3x3", dtype, Pin, col, Nimg, Ti, Fi)
        gemm(dtype, 0, 0, M, 64, 576, col, 576, Wf, 576, Y, 64, bias=bias, out_f32=int(ad == N.F32))
    Pout = _empty((Nimg, Ti // 2, Fi // 2, 64), td, like=Pin)
    if train and fused and SYNC_BN is None:
        # statistics -> scale / shift inside the pooling launch (no bn_finalize node between the convolution and the pooling)
        st = _empty((4, 64), torch.float32, like=Pin)
        N.call("bn_relu_pool_train_fwd", dtype, ad, Y, sq[0, 0], sq[0, 1], rep, float(M), gamma, beta, rm if update_running else None,
               rv if update_running else None, BN_MOMENTUM, BN_EPS, Pout, st[0], st[1], st[2], st[3], Nimg, Ti, Fi)
        return Y, Pout, (st[0], st[1], st[2], st[3]), Wf, Wd
    if train and fused:
        st = _empty((4, 64), torch.float32, like=Pin)
        sq = sq[0]
        SYNC_BN.all_reduce(sq)
        N.call("bn_finalize", sq[0], sq[1], 1, float(M * SYNC_BN.world), 64, gamma, beta, rm if update_running else None,
               rv if update_running else None, BN_MOMENTUM, BN_EPS, st[0], st[1], st[2], st[3])
        scale, shift, mean, rstd = st[0], st[1], st[2], st[3]
    elif train:
        scale, shift, mean, rstd = _bn_train(ad, Y, M, 64, gamma, beta, rm, rv, update_running)
    else:
        scale, shift = _bn_eval(Y, 64, gamma, beta, rm, rv)
        mean = rstd = None
    N.call("bn_relu_pool_fwd", dtype, ad, Y, scale, shift, Pout, Nimg, Ti, Fi)
    return Y, Pout, (scale, shift, mean, rstd), Wf, Wd


def _col_buffer(dtype, Nimg, T1, F1, like):
    """im2col scratch of the fp32 path (the bf16 path convolves implicitly)."""
    if dtype == N.BF16 and F1 in (16, 32):
        return None
    return _empty((Nimg * T1 * F1, 576), N.torch_dtype(dtype), like=like)


def encoder_forward(P, x, dtype, keep=None, p_drop=0.3, train=True, update_running=True, want_layers=True, Wc=None,
                    layer_out=None, before_fc=None):
    """P: dict of fp32 device tensors with the reference's state_dict keys (features_1.0.weight, ...).
    x [N,1,F,T] fp32.  keep: uint8 [N*T3, d] dropout keep-mask or None (no dropout).
    Wc: optional {"fc.0.weight", "fc.3.weight"} already in the activation dtype (flat shadow); layer_out: optional
    (x1, x2, x3) output buffers [N, F*64] (e.g. halves of the stacked projector inputs).
    Returns (x1, x2, x3, H2 [N,T3,d]) in the activation dtype and the ctx for the backward."""
    Nimg, _, F, T = x.shape
    assert F == 64, "the HIP encoder is built for n_mels = 64 (the reference's only setting)"
    td = N.torch_dtype(dtype)
    c = EncoderCtx()
    c.P = P
    c.dtype, c.N, c.F, c.T, c.train, c.p_drop = dtype, Nimg, F, T, train, p_drop
    img = x.reshape(Nimg, F, T).contiguous()
    c.img = img
    w1 = P["features_1.0.weight"].reshape(64, 9)
    b1 = P["features_1.0.bias"]
    g1, be1 = P["features_1.1.weight"], P["features_1.1.bias"]
    if train:
        c.mom1 = ARENA.scratch((16 * 54,), torch.float64, x)         # 16 replicas of the 54 tap moments; totals end up in [0:54]
        c.sc1, c.sh1, c.mean1, c.rstd1 = (_empty((64,), torch.float32, like=x) for _ in range(4))
        rm1 = P["features_1.1.running_mean"] if update_running else None
        rv1 = P["features_1.1.running_var"] if update_running else None
        if SYNC_BN is None:
            N.call("conv1_stats", img, Nimg, F, T, w1, b1, g1, be1, rm1, rv1, BN_MOMENTUM, BN_EPS, c.mom1, c.sc1, c.sh1, c.mean1, c.rstd1)
        else:                                     # this rank's tap moments stay in c.mom1 (the backward needs the local sums)
            N.call("conv1_moments", img, Nimg, F, T, c.mom1)
            mom_g = SYNC_BN.all_reduce(c.mom1[:54].clone())
            N.call("conv1_finalize", mom_g, w1, b1, g1, be1, rm1, rv1, BN_MOMENTUM, BN_EPS, float(Nimg * F * T * SYNC_BN.world),
                   c.sc1, c.sh1, c.mean1, c.rstd1)
    else:
        c.sc1, c.sh1 = _bn_eval(x, 64, g1, be1, P["features_1.1.running_mean"], P["features_1.1.running_var"])
    T1, F1 = T // 2, F // 2
    c.P1 = _empty((Nimg, T1, F1, 64), td, like=x)
    # bf16: the stem conv is a bf16 MFMA; bf16_hp keeps the fp32 VALU convolution (exact pooling arg-max) with bf16 output
    c.stem_mfma = dtype == N.BF16 and not HP and not _STEM_VALU
    x1 = x2 = x3 = None
    if want_layers:
        x1, x2, x3 = layer_out if layer_out is not None else (_empty((Nimg, f * 64), td, like=x) for f in (F // 2, F // 4, F // 8))
    # the MFMA stem also leaves x_1 (mean over time of its pooled output) when an fp32 buffer is wanted for it
    x1_fused = c.stem_mfma and x1 is not None and x1.dtype == torch.float32 and x1.is_contiguous()
    x1_parts = _empty((4, Nimg, (F // 2) * 64), torch.float32, like=x) if x1_fused else None        # AUDIOSSL_CONV1_XL_PARTS
    N.call("conv1_fwd", dtype if (dtype == N.F32 or c.stem_mfma) else 2, img, Nimg, F, T, w1, b1, c.sc1, c.sh1, c.P1, x1_parts)
    col = _col_buffer(dtype, Nimg, T1, F1, x)
    bn2 = (P["features_2.1.weight"], P["features_2.1.bias"], P["features_2.1.running_mean"], P["features_2.1.running_var"])
    packs = [_empty((64, 576), td, like=x) for _ in range(4)]         # both layers' (forward, data-gradient) layouts: one launch
    N.call("pack_conv_w2", dtype, P["features_2.0.weight"], packs[0], packs[1], P["features_3.0.weight"], packs[2], packs[3])
    c.Y2, c.P2, c.st2, c.W2f, c.W2d = _conv_block_fwd(dtype, c.P1, Nimg, T1, F1, P["features_2.0.weight"],
                                                       P["features_2.0.bias"], bn2, train, update_running, col, packed=packs[:2])
    T2, F2 = T1 // 2, F1 // 2
    bn3 = (P["features_3.1.weight"], P["features_3.1.bias"], P["features_3.1.running_mean"], P["features_3.1.running_var"])
    c.Y3, c.P3, c.st3, c.W3f, c.W3d = _conv_block_fwd(dtype, c.P2, Nimg, T2, F2, P["features_3.0.weight"],
                                                       P["features_3.0.bias"], bn3, train, update_running, col, packed=packs[2:])
    del col
    T3, F3 = T2 // 2, F2 // 2
    c.dims = (T1, F1, T2, F2, T3, F3)
    if want_layers:
        o32 = int(x1.dtype == torch.float32)
        N.call("tmean3_fwd", dtype, o32, None if x1_fused else c.P1, x1, x1_parts, T1, F1, c.P2, x2, T2, F2, c.P3, x3, T3, F3, Nimg)
    d = P["fc.0.weight"].shape[0]
    kin = F3 * 64
    M = Nimg * T3
    c.d, c.kin, c.M = d, kin, M
    if before_fc is not None:
        before_fc()                      # e.g. join the stream that refreshes the bf16 weight shadow: first use is below
    if Wc is not None:
        c.fw1, c.fw2 = Wc["fc.0.weight"], Wc["fc.3.weight"]
    else:
        c.fw1, c.fw2 = cast(dtype, P["fc.0.weight"]), cast(dtype, P["fc.3.weight"])
    if keep is not None and not isinstance(keep, torch.Tensor) and (dtype != N.BF16 or not train):
        keep = keep.materialise()        # a virtual mask (encoder.audiontt.VirtualKeep) outside the bf16 training path: write it out
    c.keep = keep if train else None
    scale = 1.0 / (1.0 - p_drop) if c.keep is not None else 1.0
    if c.keep is not None and not isinstance(c.keep, torch.Tensor):
        # Linear -> ReLU -> Dropout with the mask drawn by the GEMM for its own elements: nothing written, nothing read back
        vk = c.keep
        c.H1 = torch.empty(M, d, dtype=torch.bfloat16, device=c.P3.device)
        N.call("gemm_dropout", 0, 0, M, d, kin, 1.0, c.P3, kin, c.fw1, kin, c.H1, d, P["fc.0.bias"], 1, vk.seed, float(vk.p), vk.counter,
               float(scale))
    else:
        c.H1 = linear_fwd(dtype, c.P3, c.fw1, M, d, kin, bias=P["fc.0.bias"], relu=1, keep=c.keep, keep_scale=scale)
    c.H2 = linear_fwd(dtype, c.H1, c.fw2, M, d, d, bias=P["fc.3.bias"], relu=1)
    return x1, x2, x3, c.H2.view(Nimg, T3, d), c


def _conv_block_bwd(dtype, Y, dP, dxl, st, Nimg, Ti, Fi, Pin, Wd, G_w, G_gamma, G_beta, need_dx, col, keep, dx_bf16=False, Pout=None):
    """BN/ReLU/pool backward + conv wgrad (+ dgrad).  Returns dPin or None.  `keep`: list that holds the tensors the
    side-stream weight gradient still reads until the caller joins the streams."""
    td = N.torch_dtype(dtype)
    scale, shift, mean, rstd = st
    M = Nimg * Ti * Fi
    dY = torch.empty((M, 64), dtype=td, device=Y.device)
    stat = ARENA.scratch((33 * 128,), torch.float32, Y)
    yd = N.F32 if Y.dtype == torch.float32 else dtype
    if SYNC_BN is None and Pout is not None and _BN_STATS_P and Pout.dtype == td:
        # dbeta / dgamma from the block's pooled output (26 MB) instead of a first sweep over Y (105 MB at B = 512, block 2)
        N.call("bn_relu_pool_bwd_p", dtype, yd, GD, Y, Pout, dP, dxl, scale, shift, mean, rstd, stat, dY, G_gamma, G_beta, Nimg, Ti, Fi)
    elif SYNC_BN is None:
        N.call("bn_relu_pool_bwd", dtype, yd, GD, Y, dP, dxl, scale, shift, mean, rstd, stat, dY, G_gamma, G_beta, Nimg, Ti, Fi)
    else:
        N.call("bn_relu_pool_bwd_stats", dtype, yd, GD, Y, dP, dxl, scale, shift, mean, rstd, stat, Nimg, Ti, Fi)
        gstat = SYNC_BN.all_reduce(stat[:128].clone())
        N.call("bn_relu_pool_bwd_apply", dtype, yd, GD, Y, dP, dxl, scale, shift, mean, rstd, stat, gstat, float(M * SYNC_BN.world), dY,
               G_gamma, G_beta, Nimg, Ti, Fi)
    # wgrad: dWp[co][tap*64+ci] = sum_pix dY[pix][co] * Pin[pix + off(tap)][ci]
    fused = dtype == N.BF16 and Fi in (16, 32)
    if fused:
        def wg():
            dWp = ARENA.zeros((64, 576), torch.float32, device=Y.device)
            ws = _wgrad_workspace(Y.device)
            N.call("conv3x3_wgrad", dY, Pin, dWp, ws, ws.numel(), Nimg, Ti, Fi)
            N.call("unpack_conv_dw", dWp, G_w)
            return dWp
        keep.append((dY, WGRAD.run(Y.device, wg)))             # off the critical path
    else:
        dWp = ARENA.zeros((64, 576), torch.float32, device=Y.device)
        N.call("im2col3x3", dtype, Pin, col, Nimg, Ti, Fi)
        gemm(dtype, 1, 1, 64, 576, M, dY, 64, col, 576, dWp, 576, out_f32=1, atomic=1, ksplit=_ksplit(64, 576, M, 1024))
        N.call("unpack_conv_dw", dWp, G_w)
    if not need_dx:
        return None
    # dgrad: dPin[pix][ci] = sum_{tap,co} dY[pix + off(tap)][co] * W[co][ci][8 - tap]
    # feeds the previous block's BN backward: fp32 - except into the MFMA stem backward, which takes this (the largest gradient
    # tensor of the step: 210 MB in fp32 at B = 512) in bf16: its sums run over 3.3 M pixels, the rounding averages out
    dx_bf16 = dx_bf16 and fused
    dPin = _empty((Nimg, Ti, Fi, 64), torch.bfloat16 if dx_bf16 else torch.float32, like=Y)
    if fused:
        N.call("conv3x3_fwd", dY, Wd, None, dPin, 0 if dx_bf16 else 1, None, None, 1, Nimg, Ti, Fi)
    else:
        N.call("im2col3x3", dtype, dY, col, Nimg, Ti, Fi)
        gemm(dtype, 0, 0, M, 64, 576, col, 576, Wd, 576, dPin, 64)
    return dPin


def encoder_backward(c, G, dA2=None, dH2=None, dx1=None, dx2=None, dx3=None, dx_late=None, grads_zero=False):
    """Accumulate parameter gradients of one encoder_forward call into G (dict keyed like P, fp32).
    dA2: grad w.r.t. the pre-ReLU output of fc.3 (already ReLU-gated), or dH2: grad w.r.t. H2.
    dx1..dx3: grads w.r.t. the temporal layer means (fp32) or None; dx_late: callable -> (dx1, dx2, dx3), called after the
    fully connected layers' backward has been issued (it may order the stream after the producer of those gradients)."""
    dtype, Nimg, M, d, kin = c.dtype, c.N, c.M, c.d, c.kin
    T1, F1, T2, F2, T3, F3 = c.dims
    td = N.torch_dtype(dtype)
    assert c.train, "backward needs a train-mode forward (batch statistics)"
    if dA2 is None:
        dA2 = _empty((M, d), td, like=c.H2)
        N.call("relu_bwd", dtype, dH2.contiguous(), c.H2, dA2, M * d)
    dev = c.H2.device
    # The data-gradient chain (dA2 -> dA1 -> dP3 -> dY3 -> ...) is the critical path; every weight / bias gradient
    # only hangs off it, so those launches go to a second stream and overlap the chain.
    # fc.3
    def w2():
        linear_bwd_w(dtype, dA2, c.H1, G["fc.3.weight"], M, d, d, dw_zero=grads_zero)
        colsum_add(dtype, dA2, M, d, G["fc.3.bias"])
    WGRAD.run(dev, w2)
    scale = 1.0 / (1.0 - c.p_drop) if c.keep is not None else 1.0
    dA1 = linear_bwd_x(dtype, dA2, c.fw2, M, d, d, alpha=scale, gate=c.H1)
    # fc.0
    def w1():
        linear_bwd_w(dtype, dA1, c.P3, G["fc.0.weight"], M, d, kin, dw_zero=grads_zero)
        colsum_add(dtype, dA1, M, d, G["fc.0.bias"])
    WGRAD.run(dev, w1)
    dP3 = linear_bwd_x(dtype, dA1, c.fw1, M, d, kin, out_f32=1)
    if dx_late is not None:
        dx1, dx2, dx3 = dx_late()
    col = _col_buffer(dtype, Nimg, T1, F1, c.H2)
    keep = []
    dP2 = _conv_block_bwd(dtype, c.Y3, dP3, dx3, c.st3, Nimg, T2, F2, c.P2, c.W3d, G["features_3.0.weight"],
                          G["features_3.1.weight"], G["features_3.1.bias"], True, col, keep, Pout=c.P3)
    dP1 = _conv_block_bwd(dtype, c.Y2, dP2, dx2, c.st2, Nimg, T1, F1, c.P1, c.W2d, G["features_2.0.weight"],
                          G["features_2.1.weight"], G["features_2.1.bias"], True, col, keep, dx_bf16=bool(c.stem_mfma) and _DP1_BF16,
                          Pout=c.P2)
    gd1 = N.BF16 if dP1.dtype == torch.bfloat16 else GD
    acc = ARENA.scratch((32 * 64 * 11,), torch.float32, c.H2)
    P = c.P
    w1, b1, g1 = P["features_1.0.weight"].reshape(64, 9), P["features_1.0.bias"], P["features_1.1.weight"]
    gw, gb = G["features_1.0.weight"].view(64, 9), G["features_1.0.bias"]
    if SYNC_BN is None:
        N.call("conv1_bwd", gd1, int(c.stem_mfma), c.img, Nimg, c.F, c.T, w1, b1, g1, c.sc1, c.sh1, c.mean1, c.rstd1, c.mom1, dP1, dx1,
               acc, gw, gb, G["features_1.1.weight"], G["features_1.1.bias"])
    else:
        lstat = _empty((128,), torch.float32, like=c.H2)
        N.call("conv1_bwd_sums", gd1, int(c.stem_mfma), c.img, Nimg, c.F, c.T, w1, b1, g1, c.sc1, c.sh1, c.mean1, c.rstd1, c.mom1, dP1,
               dx1, acc, lstat)
        gstat = SYNC_BN.all_reduce(lstat.clone())
        N.call("conv1_bwd_finalize", acc, c.mom1, w1, b1, g1, c.mean1, c.rstd1, float(Nimg * c.F * c.T * SYNC_BN.world), gstat, gw, gb,
               G["features_1.1.weight"], G["features_1.1.bias"])
    WGRAD.join(dev)            # dA2 / dA1 / dY* stay referenced until the side stream is ordered before us


# =============================================================================================== max + mean pooling
def maxmean_forward(dtype, H2, out=None):
    """H2 [N,T3,d] -> y [N,d] = max_T + mean_T, plus the argmax needed by the backward."""
    Nimg, T3, d = H2.shape
    y = out if out is not None else _empty((Nimg, d), N.torch_dtype(dtype), like=H2)
    arg = _empty((Nimg, d), torch.uint8, like=H2)
    N.call("maxmean_fwd", dtype, int(y.dtype == torch.float32), H2, y, arg, Nimg, T3, d)
    return y, arg


def maxmean_backward(dtype, dy, arg, H2):
    """dy [N, d] fp32 -> dA2 [N*T3, d] (activation dtype: it is only an MFMA operand afterwards): gradient w.r.t. the
    pre-ReLU fc.3 output, ReLU gate fused."""
    Nimg, T3, d = H2.shape
    dA2 = _empty((Nimg * T3, d), N.torch_dtype(dtype), like=H2)
    N.call("maxmean_bwd", dtype, GD, dy, arg, H2, dA2, Nimg, T3, d)
    return dA2


# =============================================================================================== Barlow projector
class ProjCtx:
    pass


def projector_forward(PP, Y, dtype, groups, B, update_running=True, Wc=None):
    """Lin-BN-ReLU-Lin-BN-ReLU-Lin then affine-free BN on `groups` stacked batches Y [groups*B, in] (the two views share
    the weights, so they go through each GEMM together; BatchNorm statistics stay per view).
    PP keys: projector.{0,3,6}.weight, projector.{1,4}.{weight,bias,running_mean,running_var}, bn.{running_*}."""
    kin = Y.shape[1]
    c = ProjCtx()
    c.dtype, c.B, c.groups, c.kin, c.y = dtype, B, groups, kin, Y
    W = Wc if Wc is not None else tuple(cast(dtype, PP[f"projector.{i}.weight"]) for i in (0, 3, 6))
    c.W = W
    D = W[0].shape[0]
    c.D = D
    M = groups * B

    def bn(a, prefix, affine, relu, out, adt=None):
        g = PP[prefix + ".weight"] if affine else None
        b = PP[prefix + ".bias"] if affine else None
        return bn_train_apply(dtype, ad if adt is None else adt, a, B, D, g, b, PP[prefix + ".running_mean"], PP[prefix + ".running_var"],
                              update_running, groups, relu, out)
    td = N.torch_dtype(dtype)
    ad = _ad(dtype)
    c.ad = ad
    o32 = int(ad == N.F32)
    # pre-BatchNorm tensors (a1, a2, z) are fp32 GEMM outputs on the fp32 / bf16_hp paths; the normalised activations
    # (MFMA operands) are `dtype`
    c.cmean = None
    if dtype == N.BF16 and Y.dtype == torch.float32 and not HP and B <= 1024 and kin % 32 == 0 and SYNC_BN is None:
        # time-pooled post-ReLU features: |mean| >> batch-std, so a single bf16 rounding would eat the batch variation
        # that BatchNorm amplifies.  The layer is Linear(no bias) -> train-mode BatchNorm, i.e. blind to a constant per input
        # column: feed it the CENTRED features (csrc/heads.hip center_cast_kernel); the weight gradient is unchanged too.
        c.y_hi, c.cmean = center_cast(Y, groups, B)
        c.y_lo = None
        c.a1 = linear_fwd(dtype, c.y_hi, W[0], M, D, kin, out_f32=o32)
    elif dtype == N.BF16 and Y.dtype == torch.float32:
        # bf16_hp - and SyncBatchNorm, where a per-RANK column mean is not a constant of the (global) batch any more: the first GEMM
        # on hi + lo bf16 pieces of the operand (fp32 accumulate into the same output)
        c.y_hi, c.y_lo = _empty((M, kin), td, like=Y), _empty((M, kin), td, like=Y)
        N.call("split_bf16", Y, c.y_hi, c.y_lo, M * kin)
        c.a1 = torch.zeros(M, D, dtype=torch.float32, device=Y.device)
        gemm(dtype, 0, 0, M, D, kin, c.y_hi, kin, W[0], kin, c.a1, D, out_f32=1, atomic=1)
        gemm(dtype, 0, 0, M, D, kin, c.y_lo, kin, W[0], kin, c.a1, D, out_f32=1, atomic=1)
    else:
        c.y_hi, c.y_lo = Y, None
        c.a1 = linear_fwd(dtype, Y, W[0], M, D, kin, out_f32=o32)
    c.ad1 = N.F32 if c.a1.dtype == torch.float32 else ad     # the hi + lo form accumulates the first layer in fp32 whatever the path
    c.h1 = _empty((M, D), td, like=Y)
    c.st1 = bn(c.a1, "projector.1", True, 1, c.h1, adt=c.ad1)
    if c.cmean is not None and update_running:
        N.call("shift_running_mean", W[0], c.cmean, PP["projector.1.running_mean"], D, kin, groups, BN_MOMENTUM)
    c.a2 = linear_fwd(dtype, c.h1, W[1], M, D, D, out_f32=o32)
    c.h2 = _empty((M, D), td, like=Y)
    c.st2 = bn(c.a2, "projector.4", True, 1, c.h2)
    c.z = linear_fwd(dtype, c.h2, W[2], M, D, D, out_f32=o32)
    c.zn = _empty((M, D), td, like=Y)
    c.st0 = bn(c.z, "bn", False, 0, c.zn)
    return c.zn, c


def projector_backward(c, PP, G, dzn, dy_rows=None):
    """dzn [groups*B, D] (activation dtype) -> accumulates projector grads into G; returns dY for the first `dy_rows`
    rows of the stacked input (None: all rows, 0: skip)."""
    dtype, B, D, kin, groups = c.dtype, c.B, c.D, c.kin, c.groups
    M = groups * B
    td = N.torch_dtype(dtype)

    def bn_bwd(a, dh, st, relu, da, g_gamma, g_beta, adt=None):
        adt = c.ad if adt is None else adt
        """BatchNorm1d(train) backward per group (view).  SyncBatchNorm (`extras/delores-s/main.py:79` converts every BatchNorm of
        the model, the projector's included): the two means of the backward are over the GLOBAL batch of the view - this rank's
        sums are all-reduced between the statistics pass and the apply pass; the parameter gradients stay this rank's own sums
        (the data-parallel gradient all-reduce adds the ranks up afterwards)."""
        t = ARENA.scratch((2 * groups * D,), torch.float64, dzn)      # a fresh (zeroed) statistics scratch per BatchNorm backward
        if SYNC_BN is None:
            N.call("colbn_bwd", dtype, adt, GD, a, dh, *st, relu, groups, B, D, t, da, g_gamma, g_beta)
            return
        N.call("colbn_bwd_stats", dtype, adt, GD, a, dh, *st, relu, groups, B, D, t)
        tg = SYNC_BN.all_reduce(t.clone())
        N.call("colbn_bwd_apply", dtype, adt, GD, a, dh, *st, relu, groups, B, D, tg, float(B * SYNC_BN.world), da)
        if g_gamma is not None:
            for g in range(groups):              # t = [dbeta sums: groups x D][dgamma sums: groups x D]
                N.call("add_d2f", t[g * D:(g + 1) * D], g_beta, D)
                N.call("add_d2f", t[(groups + g) * D:(groups + g + 1) * D], g_gamma, D)
    dz = _empty((M, D), td, like=c.zn)
    # gradients entering a BatchNorm backward (dzn, dh2, dh1) are fp32 GEMM outputs; its outputs (dz, da2, da1) are
    # MFMA operands only and are stored in the activation dtype
    bn_bwd(c.z, dzn, c.st0, 0, dz, None, None)
    linear_bwd_w(dtype, dz, c.h2, G["projector.6.weight"], M, D, D)
    dh2 = linear_bwd_x(dtype, dz, c.W[2], M, D, D, out_f32=1)
    da2 = _empty((M, D), td, like=c.zn)
    bn_bwd(c.a2, dh2, c.st2, 1, da2, G["projector.4.weight"], G["projector.4.bias"])
    linear_bwd_w(dtype, da2, c.h1, G["projector.3.weight"], M, D, D)
    dh1 = linear_bwd_x(dtype, da2, c.W[1], M, D, D, out_f32=1)
    da1 = _empty((M, D), td, like=c.zn)
    bn_bwd(c.a1, dh1, c.st1, 1, da1, G["projector.1.weight"], G["projector.1.bias"], adt=c.ad1)
    linear_bwd_w(dtype, da1, c.y_hi, G["projector.0.weight"], M, D, kin)
    if c.y_lo is not None:
        linear_bwd_w(dtype, da1, c.y_lo, G["projector.0.weight"], M, D, kin)
    rows = M if dy_rows is None else dy_rows
    if rows == 0:
        return None
    return linear_bwd_x(dtype, da1, c.W[0], rows, D, kin, out_f32=1)


def barlow_forward_backward(PP, G, Y, dtype, lambd, scale_loss, loss_out, need_dy1=True, need_dy2=True,
                            update_running=True, all_reduce=None, global_batch=None, backward=True, Wc=None):
    """Projection.forward + its whole backward on the stacked views Y [2B, in] (rows [0,B) = view 1, [B,2B) = view 2).
    loss_out: fp32 device scalar, accumulated (+=).  all_reduce: optional callable summing the [D,D] fp32 correlation
    over ranks in place (cross-GPU Barlow, `extras/delores-s/models_byol.py:108-112`); the divisor is then
    `global_batch`.  Returns dY [rows, in] in the activation dtype: rows = 2B, B (only view 1) or None."""
    B = Y.shape[0] // 2
    zn, c = projector_forward(PP, Y, dtype, 2, B, update_running, Wc)
    D = c.D
    zn1, zn2 = zn[:B], zn[B:]
    denom = float(global_batch if global_batch else B)
    cmat = _empty((D, D), torch.float32, like=Y)
    gemm(dtype, 1, 1, D, D, B, zn1, D, zn2, D, cmat, D, alpha=1.0 / denom, out_f32=1)         # c = zn1^T zn2 / B
    if all_reduce is not None:
        all_reduce(cmat)
    coef = (lambd if lambd else 1.0) * scale_loss
    dc = _empty((D, D), N.torch_dtype(dtype), like=Y)
    N.call("barlow_loss", dtype, cmat, D, coef, 2.0 * coef / denom, dc, loss_out)
    if not backward:
        return None
    dzn = _empty((2 * B, D), torch.float32, like=Y)            # enters the affine-free BatchNorm backward: fp32
    gemm(dtype, 0, 0, B, D, D, zn2, D, dc, D, dzn[:B], D, out_f32=1)      # dzn1[b,i] = sum_j zn2[b,j] dc[i,j]
    gemm(dtype, 0, 1, B, D, D, zn1, D, dc, D, dzn[B:], D, out_f32=1)      # dzn2[b,j] = sum_i zn1[b,i] dc[i,j]
    rows = 2 * B if need_dy2 else (B if need_dy1 else 0)
    return projector_backward(c, PP, G, dzn, rows)


# =============================================================================================== several Barlow heads at once
def _harr(ctype, xs):
    """host array of scalars / device pointers for the multi-problem entry points (None -> null pointer)"""
    import ctypes
    vals = [x.data_ptr() if isinstance(x, torch.Tensor) else x for x in xs]
    return (ctype * len(vals))(*vals)


def _multi_check(*lists):
    for ts in lists:
        for t in ts:
            if t is not None and (not t.is_cuda or not t.is_contiguous()):
                raise RuntimeError("multi-problem launches need contiguous device tensors")


def gemm_multi(ta, tb, M, Nn, Ks, As, ldas, Bs, ldbs, Cs, ldc, alpha=1.0, out_f32=0, atomic=0, ksplit=1):
    """One launch for len(As) independent bf16 problems with common M, transposes and epilogue (audiossl_gemm_multi).
    Nn / ldc: one int for all problems or one per problem."""
    import ctypes
    n = len(As)
    _multi_check(As, Bs, Cs)
    Ns = [Nn] * n if isinstance(Nn, int) else list(Nn)
    ldcs = [ldc] * n if isinstance(ldc, int) else list(ldc)
    adr = ctypes.addressof
    arrs = (_harr(ctypes.c_int, Ns), _harr(ctypes.c_int, Ks), _harr(ctypes.c_void_p, As), _harr(ctypes.c_long, ldas),
            _harr(ctypes.c_void_p, Bs), _harr(ctypes.c_long, ldbs), _harr(ctypes.c_void_p, Cs), _harr(ctypes.c_long, ldcs))
    N_, K_, A_, la_, B_, lb_, C_, lc_ = arrs
    if N.PROFILE is not None:
        N.PROFILE_NOTE = float(sum(2.0 * M * nn * kk for nn, kk in zip(Ns, Ks)))        # flops of the launch (bench.py roofline)
    N.call("gemm_multi", n, ta, tb, M, adr(N_), adr(K_), float(alpha), adr(A_), adr(la_), adr(B_), adr(lb_), adr(C_), adr(lc_),
           out_f32, atomic, ksplit)


def gemm_multi_sgd(ta, tb, M, Nn, Ks, As, ldas, Bs, ldbs, Ps, Moms, Shadows, ldp, hyper, alpha=1.0):
    """`gemm_multi` whose results are weight gradients applied on the spot (audiossl_gemm_multi_sgd): Ps / Moms fp32 parameters and
    momentum buffers, Shadows the bf16 copies (or None); hyper = (lr, momentum, weight_decay, grad_scale, grad_scale_tensor)."""
    import ctypes
    n = len(As)
    _multi_check(As, Bs, Ps, Moms, [] if Shadows is None else Shadows)
    Ns = [Nn] * n if isinstance(Nn, int) else list(Nn)
    ldps = [ldp] * n if isinstance(ldp, int) else list(ldp)
    adr = ctypes.addressof
    vp = ctypes.c_void_p
    arrs = (_harr(ctypes.c_int, Ns), _harr(ctypes.c_int, Ks), _harr(vp, As), _harr(ctypes.c_long, ldas), _harr(vp, Bs),
            _harr(ctypes.c_long, ldbs), _harr(vp, Ps), _harr(vp, Moms), _harr(vp, Shadows if Shadows is not None else [None] * n),
            _harr(ctypes.c_long, ldps))
    if N.PROFILE is not None:                # (flops, algorithmic bytes: operands + parameter and momentum read / written + shadow)
        N.PROFILE_NOTE = (float(sum(2.0 * M * nn * kk for nn, kk in zip(Ns, Ks))),
                          float(sum(2.0 * kk * (M + nn) + M * nn * (16.0 + (2.0 if Shadows is not None else 0.0)) for nn, kk in zip(Ns, Ks))))
    lr, mu, wd, gs, gs_dev = hyper
    N.call("gemm_multi_sgd", n, ta, tb, M, adr(arrs[0]), adr(arrs[1]), float(alpha), adr(arrs[2]), adr(arrs[3]), adr(arrs[4]), adr(arrs[5]),
           adr(arrs[6]), adr(arrs[7]), adr(arrs[8]), adr(arrs[9]), float(lr), float(mu), float(wd), float(gs), gs_dev)


def _heads_wgrad_split(D, kmax, M, nh):
    return _ksplit(D, kmax, M, max(256 // nh, 1))


def heads_wgrads_store(D, kins, M, grads_zero=True):
    """True when every weight-gradient launch of `barlow_heads_forward_backward(..., grads_zero=True)` STORES its result (one
    unsplit writer per dW): the caller may then leave those gradients uncleared between steps (HipSGD.step_tail(stored=...))."""
    nh = len(kins)
    return bool(grads_zero) and all(_heads_wgrad_split(D, k, M, nh) == 1 for k in (D, max(kins)))


def barlow_heads_forward_backward(PPs, Gs, Ys, dtype, lambds, scale_losses, loss_outs, update_running=True, backward=True,
                                  Wcs=None, grads_zero=False, dy_ready=None, sgd=None):
    """`barlow_forward_backward` for several heads in lock-step: every step of the chain - GEMM, train-mode BatchNorm, loss -
    is ONE multi-problem launch over the heads (they differ only in the width of the first layer's input).
    Ys[h]: [2B, in_h] stacked views; returns dY_h [B, in_h] (gradient of view 1's input) per head, fp32.
    bf16 path only (not bf16_hp: split first GEMM), no cross-GPU correlation all-reduce; B <= 1024."""
    import ctypes
    nh = len(Ys)
    B = Ys[0].shape[0] // 2
    M = 2 * B
    td = N.torch_dtype(dtype)
    dev = Ys[0].device
    W = Wcs if Wcs is not None else [tuple(cast(dtype, PP[f"projector.{i}.weight"]) for i in (0, 3, 6)) for PP in PPs]
    D = W[0][0].shape[0]
    kins = [Y.shape[1] for Y in Ys]
    ad = _ad(dtype)
    if dtype != N.BF16 or ad != N.BF16 or B > 1024 or D % 32:
        raise RuntimeError("grouped Barlow heads: bf16 path with B <= 1024 only")
    H = range(nh)
    vp, fl, adr = ctypes.c_void_p, ctypes.c_float, ctypes.addressof
    new = lambda rows, cols, t: [torch.empty(rows, cols, dtype=t, device=dev) for _ in H]

    def layer(xs, ws, ks, prefix, affine, relu):
        a = new(M, D, td)
        gemm_multi(0, 0, M, D, ks, xs, ks, ws, ks, a, D)
        out, st = new(M, D, td), new(4, 2 * D, torch.float32)
        par = lambda suffix, on=True: _harr(vp, [PP[prefix + suffix] if on else None for PP in PPs])
        arrs = (_harr(vp, a), par(".weight", affine), par(".bias", affine), par(".running_mean", update_running),
                par(".running_var", update_running), _harr(vp, out), _harr(vp, st))
        _multi_check(a, out, st)
        N.call("colbn_train_fwd_multi", nh, ad, adr(arrs[0]), adr(arrs[1]), adr(arrs[2]), adr(arrs[3]), adr(arrs[4]),
               BN_MOMENTUM, BN_EPS, relu, 2, B, D, adr(arrs[5]), adr(arrs[6]))
        return a, st, out
    # centred bf16 operands of the first layer (see projector_forward); Ys arrive in fp32
    ycs = [torch.empty(M, k, dtype=torch.bfloat16, device=dev) for k in kins]
    cms = [torch.empty(2, k, dtype=torch.float32, device=dev) for k in kins]
    _multi_check(list(Ys), ycs, cms)
    if any(Y.dtype != torch.float32 or k % 32 for Y, k in zip(Ys, kins)):
        raise RuntimeError("grouped Barlow heads: fp32 inputs with width % 32 == 0")
    arrs = (_harr(vp, list(Ys)), _harr(vp, ycs), _harr(vp, cms), _harr(ctypes.c_int, kins))
    N.call("center_cast_multi", nh, adr(arrs[0]), adr(arrs[1]), adr(arrs[2]), adr(arrs[3]), 2, B)
    a1, st1, h1 = layer(ycs, [W[h][0] for h in H], kins, "projector.1", True, 1)

    def shift_running():
        # repairs the one thing that sees the centring of the first layer's input (include/audiossl_hip.h: center_cast); nothing in
        # the step reads the running mean, so with a backward the launch waits until the data-gradient chain is through (it must
        # still precede the first layer's weight gradient, which - when it carries the SGD update - rewrites the weight shadow)
        rms = [PP["projector.1.running_mean"] for PP in PPs]
        w0 = [W[h][0] for h in H]
        _multi_check(w0, rms)
        arrs = (_harr(vp, w0), _harr(vp, cms), _harr(vp, rms), _harr(ctypes.c_int, kins))
        N.call("shift_running_mean_multi", nh, adr(arrs[0]), adr(arrs[1]), adr(arrs[2]), D, adr(arrs[3]), 2, BN_MOMENTUM)
    shift_late = update_running and backward and dy_ready is not None      # only then are the weight gradients issued after the chain
    if update_running and not shift_late:
        shift_running()
    a2, st2, h2 = layer(h1, [W[h][1] for h in H], [D] * nh, "projector.4", True, 1)
    z, st0, zn = layer(h2, [W[h][2] for h in H], [D] * nh, "bn", False, 0)
    # correlation c_h = zn1^T zn2 / B, loss, dc
    dc = new(D, D, td)
    coefs = [(lambds[h] if lambds[h] else 1.0) * scale_losses[h] for h in H]
    dsc = [2.0 * c / B for c in coefs]
    if all(l.numel() == LOSS_REPLICAS for l in loss_outs) and D % 64 == 0:
        # loss and dc come out of the GEMM's epilogue; c (three D x D fp32 matrices: 50 MB written and read back) is never stored
        _multi_check([zn[h] for h in H], dc, list(loss_outs))
        arrs = (_harr(ctypes.c_int, [B] * nh), _harr(vp, [zn[h][:B] for h in H]), _harr(ctypes.c_long, [D] * nh),
                _harr(vp, [zn[h][B:] for h in H]), _harr(ctypes.c_long, [D] * nh), _harr(vp, dc), _harr(fl, coefs), _harr(fl, dsc),
                _harr(vp, list(loss_outs)))
        if N.PROFILE is not None:
            N.PROFILE_NOTE = float(nh * 2.0 * D * D * B)
        N.call("gemm_multi_barlow", nh, D, adr(arrs[0]), 1.0 / B, adr(arrs[1]), adr(arrs[2]), adr(arrs[3]), adr(arrs[4]), adr(arrs[5]),
               adr(arrs[6]), adr(arrs[7]), adr(arrs[8]))
    else:
        cm = new(D, D, torch.float32)
        gemm_multi(1, 1, D, D, [B] * nh, [zn[h][:B] for h in H], [D] * nh, [zn[h][B:] for h in H], [D] * nh, cm, D,
                   alpha=1.0 / B, out_f32=1)
        arrs = (_harr(vp, cm), _harr(fl, coefs), _harr(fl, dsc), _harr(vp, dc), _harr(vp, list(loss_outs)))
        _multi_check(list(loss_outs))
        N.call("barlow_loss_multi", nh, adr(arrs[0]), D, adr(arrs[1]), adr(arrs[2]), adr(arrs[3]), adr(arrs[4]))
    if not backward:
        return [None] * nh
    dzn = new(M, D, torch.float32)
    gemm_multi(0, 0, B, D, [D] * nh, [zn[h][B:] for h in H], [D] * nh, dc, [D] * nh, [dzn[h][:B] for h in H], D, out_f32=1)
    gemm_multi(0, 1, B, D, [D] * nh, [zn[h][:B] for h in H], [D] * nh, dc, [D] * nh, [dzn[h][B:] for h in H], D, out_f32=1)

    def bn_bwd(a, dh, st, relu, names):
        out = new(M, D, td)
        grads = lambda k: _harr(vp, [Gs[h][names[k]] if names else None for h in H])
        arrs = (_harr(vp, a), _harr(vp, dh), _harr(vp, st), _harr(vp, out), grads(0), grads(1))
        _multi_check(a, dh, st)
        N.call("colbn_bwd_multi", nh, ad, GD, adr(arrs[0]), adr(arrs[1]), adr(arrs[2]), relu, 2, B, D, adr(arrs[3]),
               adr(arrs[4]), adr(arrs[5]))
        return out

    def wgrad(dys, xs, name, ks):                        # dW_h [D, k_h] += dy_h^T x_h; one writer per dW, so no atomics unless split
        split = _heads_wgrad_split(D, max(ks), M, nh)
        if sgd is not None:
            # the optimiser's update of these weights happens in the GEMM's epilogue (HipSGD.fused_wgrad): the gradient never reaches
            # memory.  The weight shadows W[h][i] it rewrites were last read by the data-gradient launches issued before on this stream.
            if split != 1 or Wcs is None:
                raise RuntimeError("fused SGD needs unsplit weight gradients and the flat weight shadows")
            li = {"projector.0.weight": 0, "projector.3.weight": 1, "projector.6.weight": 2}[name]
            gemm_multi_sgd(1, 1, D, ks, [M] * nh, dys, [D] * nh, xs, ks, [PPs[h][name] for h in H], [sgd["momentum"][h][name] for h in H],
                           [W[h][li] for h in H], ks, sgd["hyper"])
            return
        # grads_zero: the caller guarantees the gradient buffers are zero and written once in this step, so the single writer
        # stores its result instead of reading 50 MB of zeros per launch to add to them
        gemm_multi(1, 1, D, ks, [M] * nh, dys, [D] * nh, xs, ks, [Gs[h][name] for h in H], ks, out_f32=1,
                   atomic=1 if split > 1 else (0 if grads_zero else 2), ksplit=split)

    def dgrad(dys, ws, Ks, rows):
        out = [torch.empty(rows, k, dtype=torch.float32, device=dev) for k in Ks]
        gemm_multi(0, 1, rows, Ks, [D] * nh, [dy[:rows] for dy in dys], [D] * nh, ws, Ks, out, Ks, out_f32=1)
        return out
    # dy_ready (callable): the weight gradients hang off the data-gradient chain (bn_bwd -> dgrad -> bn_bwd ...) and nobody but the
    # optimiser needs them, so they are issued AFTER the chain and dy_ready() is called in between - the caller can let another
    # stream continue from there (an event) while this one still works through the three weight-gradient launches
    later = []
    defer = (lambda fn: later.append(fn)) if dy_ready is not None else (lambda fn: fn())
    dz = bn_bwd(z, dzn, st0, 0, None)
    defer(lambda: wgrad(dz, h2, "projector.6.weight", [D] * nh))
    dh2 = dgrad(dz, [W[h][2] for h in H], [D] * nh, M)
    da2 = bn_bwd(a2, dh2, st2, 1, ("projector.4.weight", "projector.4.bias"))
    defer(lambda: wgrad(da2, h1, "projector.3.weight", [D] * nh))
    dh1 = dgrad(da2, [W[h][1] for h in H], [D] * nh, M)
    da1 = bn_bwd(a1, dh1, st1, 1, ("projector.1.weight", "projector.1.bias"))
    defer(lambda: wgrad(da1, ycs, "projector.0.weight", kins))
    dy = dgrad(da1, [W[h][0] for h in H], kins, B)        # dY for view 1 only (rows [0, B)); widths differ per head
    if dy_ready is not None:
        dy_ready()
    if shift_late:
        shift_running()
    for fn in later:
        fn()
    return dy


# =============================================================================================== MoCo head
def moco_forward_backward(dtype, q, k, queue, queue_shadow, temperature, loss_out, backward=True):
    """q, k: fp32 [B,dim] pre-normalisation embeddings (k carries no gradient).  queue fp32 [dim,K]; queue_shadow the
    same in the activation dtype.  loss_out += CE(logits, 0).  Returns (dq in the activation dtype or None, kn fp32)."""
    B, dim = q.shape
    K = queue.shape[1]
    td = N.torch_dtype(dtype)
    qn, kn = _empty((B, dim), td, like=q), _empty((B, dim), td, like=q)
    qn32, kn32 = _empty((B, dim), torch.float32, like=q), _empty((B, dim), torch.float32, like=q)
    qinv, kinv = _empty((B,), torch.float32, like=q), _empty((B,), torch.float32, like=q)
    lpos = _empty((B,), torch.float32, like=q)
    N.call("moco_prep", dtype, q, k, B, dim, 1.0 / temperature, qn, qn32, qinv, kn, kn32, kinv, lpos)
    lse = _empty((B,), torch.float32, like=q)
    dlpos = _empty((B,), torch.float32, like=q)
    Pm = None
    if dtype == N.BF16 and K % 8 == 0 and dim % 8 == 0:
        # the [B, K] logits never reach memory: the logits GEMM reduces its own tiles to per-slab (max, sum exp) pairs
        # (4 MB instead of 134 MB written + read at B = 512, K = 65,536), and the backward recomputes them straight into
        # P = softmax * gscale (bf16), the operand of the dq GEMM
        nslot = (K + 63) // 64
        part = _empty((B, nslot, 2), torch.float32, like=q)
        N.call("moco_logits", 1, qn, queue_shadow, B, K, dim, 1.0 / temperature, part, None, 0.0, None)
        N.call("moco_lse_merge", lpos, part, B, nslot, 1.0 / (B * temperature), lse, loss_out, dlpos)
        if not backward:
            return None, kn32
        Pm = _empty((B, K), td, like=q)
        N.call("moco_logits", 2, qn, queue_shadow, B, K, dim, 1.0 / temperature, None, lse, 1.0 / (B * temperature), Pm)
    else:
        lneg = _empty((B, K), torch.float32, like=q)
        gemm(dtype, 0, 1, B, K, dim, qn, dim, queue_shadow, K, lneg, K, alpha=1.0 / temperature, out_f32=1)
        N.call("moco_ce_fwd", lpos, lneg, B, K, lse, loss_out)
        if not backward:
            return None, kn32
        Pm = _empty((B, K), td, like=q)
        N.call("moco_ce_bwd", dtype, lpos, lneg, lse, B, K, 1.0 / (B * temperature), Pm, dlpos)
    dqn = ARENA.zeros((B, dim), torch.float32, device=q.device)
    gemm(dtype, 0, 0, B, dim, K, Pm, K, queue_shadow, K, dqn, dim, out_f32=1, atomic=1, ksplit=_ksplit(B, dim, K, 256))
    dq = _empty((B, dim), td, like=q)
    N.call("l2norm_bwd", dtype, dqn, dlpos, kn32, qn32, qinv, B, dim, dq)
    return dq, kn32
