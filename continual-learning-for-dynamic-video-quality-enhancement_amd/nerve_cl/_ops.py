"""Differentiable NHWC building blocks on libnvq (internal).

Each class below is one layer type of the reference (nerve_cl/models/layers/efficient_layers.py,
nerve_cl/models/frame_recovery.py) as a ``torch.autograd.Function`` whose forward AND backward are libnvq kernel launches;
autograd only chains them.  They are what ``FrameRecoveryNet`` is assembled from and what gives the stand-alone layer
modules (``DepthwiseSeparableConv``, ``PixelShuffleUpsampler``, ``ResidualBlock``, ``CBAM``, ``TemporalConv3D``) a working
``forward`` on HIP tensors.  (``SuperResolutionNet`` does not use them: its schedule is hand-fused in ``_engine.py``.)

Convention: activations are ``[N, H, W, ld]`` (NHWC), fp32 with ``ld = pad4(C)`` or - the storage type of
FrameRecoveryNet's high-resolution stages in the bf16 mode - bf16 with ``ld = pad8(C)``; channels ``[C, ld)`` are zero.
An op's output has its input's storage type unless it takes an ``out_dtype``; the attention / fusion ops are fp32 only.
There is no CPU path: every function raises on non-HIP tensors.
"""
from __future__ import annotations

from typing import Optional

import torch

from nerve_cl import _engine
from nerve_cl import _nvq as K
from nerve_cl._nvq import Sl, check, lib, pad4, ptr, stream

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _new(like: torch.Tensor, *shape, dtype=torch.float32, zero=False) -> torch.Tensor:
    return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=like.device)


def _ws(t: torch.Tensor) -> torch.Tensor:
    return _engine.workspace(t.device)


def padc(c: int, dtype) -> int:
    """stored channel count of a C-channel activation: 16-byte multiples"""
    return (c + 7) // 8 * 8 if dtype == torch.bfloat16 else pad4(c)


def _bf(t: torch.Tensor) -> int:
    return int(t.dtype == torch.bfloat16)


def cast_slice_(dst: torch.Tensor, src: torch.Tensor, C: int, dst_coff: int = 0, src_coff: int = 0, alpha: float = 1.0,
                accumulate: bool = False) -> None:
    """dst[..., dst_coff:+C] (+)= alpha * src[..., src_coff:+C]; either side fp32 or bf16"""
    npix = dst.numel() // dst.shape[-1]
    check(lib().nvq_cast_slice(ptr(dst), dst.shape[-1], dst_coff, _bf(dst), ptr(src), src.shape[-1], src_coff, _bf(src), C, npix,
                               alpha, int(accumulate), stream()), "nvq_cast_slice")


class Cast(torch.autograd.Function):
    """storage-type change of an activation (bf16 <-> fp32), channel padding re-sized"""

    @staticmethod
    def forward(ctx, x, dtype, C: int):
        ctx.src_dtype, ctx.C = x.dtype, C
        if x.dtype == dtype:
            return x.view_as(x)
        N, H, W, _ = x.shape
        y = _new(x, N, H, W, padc(C, dtype), dtype=dtype, zero=padc(C, dtype) > pad4(C))
        cast_slice_(y, x, pad4(C))
        return y

    @staticmethod
    def backward(ctx, dy):
        return Cast.apply(dy.contiguous(), ctx.src_dtype, ctx.C), None, None


# ----------------------------------------------------------------------------- layout
def nchw_to_nhwc_(src: torch.Tensor, n_stride: int, N: int, C: int, H: int, W: int, dst: torch.Tensor, coff: int = 0,
                  czero: Optional[int] = None, src_offset: int = 0) -> None:
    """dst[..., coff:coff+C] <- N images of an NCHW tensor (src_offset / n_stride in elements pick e.g. frame t of a clip)."""
    K.require_device(src, "input")
    check(lib().nvq_nchw_to_nhwc(ptr(src, src_offset), n_stride, N, C, H, W, ptr(dst), dst.shape[-1], coff,
                                 C if czero is None else czero, stream()), "nvq_nchw_to_nhwc")


class ToNHWC(torch.autograd.Function):
    """(N,C,H,W) contiguous fp32 -> [N,H,W,pad4(C)]"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        N, C, H, W = x.shape
        out = _new(x, N, H, W, pad4(C))
        nchw_to_nhwc_(x, C * H * W, N, C, H, W, out, 0, pad4(C))
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dy):
        return ToNCHW.apply(dy, ctx.C)


class ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C):
        N, H, W, ld = x.shape
        out = _new(x, N, C, H, W)
        check(lib().nvq_nhwc_to_nchw(ptr(x), ld, 0, N, C, H, W, ptr(out), C * H * W, stream()), "nvq_nhwc_to_nchw")
        return out

    @staticmethod
    def backward(ctx, dy):
        return ToNHWC.apply(dy), None


# ----------------------------------------------------------------------------- convolutions
def _wgrad(x, Ci: int, g, Co: int, wshape, has_bias: bool, k: int, math: int):
    """weight (and bias) gradient of a stride-1 conv: x [.., ld], g = dy [.., ldg] with Co real channels.  A bf16 dy is
    handed over with its whole (8-channel aligned, zero padded) width - the kernels want 16-byte channel groups - and the
    first Co rows of the result are returned."""
    Cg = g.shape[-1] if g.dtype == torch.bfloat16 else Co
    full = _new(x, Cg, *wshape[1:])
    db = _new(x, Cg) if has_bias else None
    K.conv_wgrad(Sl(x), Ci, Sl(g, Cg), full, db, _ws(x), k, math=math)
    return full[:Co], (db[:Co] if has_bias else None)


class Conv(torch.autograd.Function):
    """nn.Conv2d(k in {1,3}, stride 1, 'same' padding) (+ bias) (+ ReLU) through the implicit-GEMM MFMA kernels.
    weight [Co, Ci, k, k] with Ci <= x.ld; output [N,H,W,pad4(Co)]."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool, math: int, out_dtype=None):
        N, H, W, ld = x.shape
        Co, Ci, k, _ = weight.shape
        dt = x.dtype if out_dtype is None else out_dtype
        assert Ci <= ld and ld % 4 == 0 and not (relu and dt != torch.float32)
        y = _new(x, N, H, W, padc(Co, dt), dtype=dt)
        wp = K.conv_pack(weight, False, ld, math=math)
        K.conv_forward(Sl(x), wp, bias, Sl(y, Co), k, relu=relu, cout_store=y.shape[-1], math=math)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu, ctx.math, ctx.has_bias = relu, math, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        Co, Ci, k, _ = weight.shape
        ld = x.shape[-1]
        dy = dy.contiguous()
        if ctx.relu:                                     # gradient w.r.t. the pre-activation
            g = torch.empty_like(dy)
            K.axpy_slice(Sl(g), Sl(dy), 1.0, accumulate=False, mask=Sl(y))
        else:
            g = dy
        dw, db = _wgrad(x, Ci, g, Co, weight.shape, ctx.has_bias, k, ctx.math)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wt = K.conv_pack(weight, True, g.shape[-1], Ci, math=ctx.math)
            K.conv_forward(Sl(g), wt, None, Sl(dx, Ci), k, cout_store=ld, math=ctx.math)
        return dx, dw, db, None, None, None


class DwConv(torch.autograd.Function):
    """depthwise 3x3, stride 1, padding 1, no bias (efficient_layers.py:38-46); weight [C,1,3,3]"""

    @staticmethod
    def forward(ctx, x, weight):
        y = torch.empty_like(x)
        K.dwconv_forward(x, weight, y)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dw = torch.empty_like(weight)
        K.dwconv_wgrad(x, dy, dw, _ws(x))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            K.dwconv_forward(dy, weight, dx, flip=True)
        return dx, dw


class TemporalConv(torch.autograd.Function):
    """nn.Conv3d(Ci, Co, (3,1,1), padding (1,0,0), bias=False) (efficient_layers.py:271-278) on a time-major batch
    x [T*B, H, W, ld]: out[t] = W0 x[t-1] + W1 x[t] + W2 x[t+1] as three accumulating 1x1 convolutions over shifted
    image ranges (zero padding in time = the missing terms at t = 0 and t = T-1)."""

    @staticmethod
    def forward(ctx, x, weight, T: int, math: int):
        NB, H, W, ld = x.shape
        B = NB // T
        Co, Ci = weight.shape[:2]
        taps = _new(x, 3, Co, Ci, 1, 1)
        check(lib().nvq_tconv_relayout(ptr(weight), ptr(taps), Co, Ci, 1, stream()), "nvq_tconv_relayout")
        y = _new(x, NB, H, W, padc(Co, x.dtype), dtype=x.dtype)
        xs, ys = Sl(x), Sl(y, Co)
        packs = [K.conv_pack(taps[k], False, ld, math=math) for k in range(3)]
        K.conv_forward(xs, packs[1], None, ys, 1, cout_store=y.shape[-1], math=math)
        if T > 1:
            K.conv_forward(xs.images(0, NB - B), packs[0], None, ys.images(B, NB), 1, accumulate=True, cout_store=y.shape[-1],
                           math=math)
            K.conv_forward(xs.images(B, NB), packs[2], None, ys.images(0, NB - B), 1, accumulate=True, cout_store=y.shape[-1],
                           math=math)
        ctx.save_for_backward(x, taps)
        ctx.T, ctx.math, ctx.wshape = T, math, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, taps = ctx.saved_tensors
        T, math = ctx.T, ctx.math
        NB, H, W, ld = x.shape
        B = NB // T
        Co, Ci = taps.shape[1:3]
        dy = dy.contiguous()
        assert dy.dtype == torch.float32 or Co % 8 == 0
        xs, gs = Sl(x), Sl(dy, Co)
        dtaps = _new(x, 3, Co, Ci, 1, 1, zero=(T == 1))
        ws = _ws(x)
        K.conv_wgrad(xs, Ci, gs, dtaps[1], None, ws, 1, math=math)
        if T > 1:
            K.conv_wgrad(xs.images(0, NB - B), Ci, gs.images(B, NB), dtaps[0], None, ws, 1, math=math)
            K.conv_wgrad(xs.images(B, NB), Ci, gs.images(0, NB - B), dtaps[2], None, ws, 1, math=math)
        dw = _new(x, *ctx.wshape)
        check(lib().nvq_tconv_relayout(ptr(dtaps), ptr(dw), Co, Ci, 0, stream()), "nvq_tconv_relayout")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gfull = Sl(dy)
            tp = [K.conv_pack(taps[k], True, dy.shape[-1], Ci, math=math) for k in range(3)]
            dxs = Sl(dx, Ci)
            K.conv_forward(gfull, tp[1], None, dxs, 1, cout_store=ld, math=math)
            if T > 1:
                # x[t] fed out[t+1] through W0 and out[t-1] through W2
                K.conv_forward(gfull.images(B, NB), tp[0], None, dxs.images(0, NB - B), 1, accumulate=True, cout_store=ld, math=math)
                K.conv_forward(gfull.images(0, NB - B), tp[2], None, dxs.images(B, NB), 1, accumulate=True, cout_store=ld, math=math)
        return dx, dw, None, None


# ----------------------------------------------------------------------------- (2+1)D convolutions, time-in-channels layout
# x_tc: [B, H, W, T * Cp] - frame t occupies channels [t * Cp, t * Cp + C), Cp = padc(C, dtype).  BatchNorm3d over such a
# tensor is BatchNorm over its view [B, H, W * T, Cp]; spatial pooling and element-wise ops do not care.
class SpatialConvTC(torch.autograd.Function):
    """nn.Conv3d(Ci, Co, (1,3,3), padding (0,1,1), bias=False) (efficient_layers.py:259-269): the same 3x3 convolution for
    every frame, reading / writing the frames' channel slices in place."""

    @staticmethod
    def forward(ctx, x, weight, T: int, math: int, out_dtype=None):
        B, H, W, ld = x.shape
        Cpi = ld // T
        Co, Ci = weight.shape[:2]
        dt = x.dtype if out_dtype is None else out_dtype
        Cpo = padc(Co, dt)
        y = _new(x, B, H, W, T * Cpo, dtype=dt)
        wp = K.conv_pack(weight, False, Cpi, math=math)
        for t in range(T):
            K.conv_forward(Sl(x, Cpi, t * Cpi), wp, None, Sl(y, Co, t * Cpo), 3, cout_store=Cpo, math=math)
        ctx.save_for_backward(x, weight)
        ctx.T, ctx.math = T, math
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        T, math = ctx.T, ctx.math
        B, H, W, ld = x.shape
        Cpi, Cpo = ld // T, dy.shape[-1] // T
        Co, Ci = weight.shape[:2]
        dy = dy.contiguous()
        Cg = Cpo if dy.dtype == torch.bfloat16 else Co
        full = _new(x, Cg, *weight.shape[1:])
        ws = _ws(x)
        for t in range(T):
            K.conv_wgrad(Sl(x, Cpi, t * Cpi), Ci, Sl(dy, Cg, t * Cpo), full, None, ws, 3, accumulate=t > 0, math=math)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wt = K.conv_pack(weight, True, Cpo, Ci, math=math)
            for t in range(T):
                K.conv_forward(Sl(dy, Cpo, t * Cpo), wt, None, Sl(dx, Ci, t * Cpi), 3, cout_store=Cpi, math=math)
        return dx, full[:Co], None, None, None


def _tconv_cat(weight, Cp: int, k0: int, nk: int, transpose: bool) -> torch.Tensor:
    Co, Ci = weight.shape[:2]
    rows = Ci if transpose else Co
    out = _new(weight, rows, nk * Cp, 1, 1)
    check(lib().nvq_tconv_cat(ptr(weight), Co, Ci, Cp, k0, nk, int(transpose), ptr(out), stream()), "nvq_tconv_cat")
    return out


class TemporalConvTC(torch.autograd.Function):
    """nn.Conv3d(Ci, Co, (3,1,1), padding (1,0,0), bias=False) (efficient_layers.py:271-278) in the time-in-channels layout:
    output frame t = one 1x1 convolution over the contiguous channels of frames t-1..t+1 (fewer at the two ends) with the
    tap weights side by side - every activation is read and written once, nothing is accumulated in memory.  T >= 2."""

    @staticmethod
    def forward(ctx, x, weight, T: int, math: int):
        B, H, W, ld = x.shape
        assert T >= 2 and ld % T == 0
        Cpi = ld // T
        Co, Ci = weight.shape[:2]
        Cpo = padc(Co, x.dtype)
        y = _new(x, B, H, W, T * Cpo, dtype=x.dtype)
        forms = {"first": (1, 2), "mid": (0, 3), "last": (0, 2)}          # (first tap, taps) of frame 0 / inner / T-1
        packs = {k: K.conv_pack(_tconv_cat(weight, Cpi, k0, nk, False), False, nk * Cpi, math=math)
                 for k, (k0, nk) in forms.items() if k != "mid" or T > 2}
        for t in range(T):
            lo, hi = max(t - 1, 0), min(t + 1, T - 1)
            form = "first" if t == 0 else "last" if t == T - 1 else "mid"
            K.conv_forward(Sl(x, (hi - lo + 1) * Cpi, lo * Cpi), packs[form], None, Sl(y, Co, t * Cpo), 1, cout_store=Cpo, math=math)
        ctx.save_for_backward(x, weight)
        ctx.T, ctx.math = T, math
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        T, math = ctx.T, ctx.math
        B, H, W, ld = x.shape
        Cpi, Cpo = ld // T, dy.shape[-1] // T
        Co, Ci = weight.shape[:2]
        dy = dy.contiguous()
        Cg = Cpo if dy.dtype == torch.bfloat16 else Co
        ws = _ws(x)
        # weight gradient: one launch per frame into the block of its form; the inner frames accumulate into "mid"
        g = {"first": _new(x, Cg, 2 * Cpi, 1, 1), "last": _new(x, Cg, 2 * Cpi, 1, 1),
             "mid": _new(x, Cg, 3 * Cpi, 1, 1) if T > 2 else None}
        seen = set()
        for t in range(T):
            lo, hi = max(t - 1, 0), min(t + 1, T - 1)
            form = "first" if t == 0 else "last" if t == T - 1 else "mid"
            nk = hi - lo + 1
            K.conv_wgrad(Sl(x, nk * Cpi, lo * Cpi), nk * Cpi, Sl(dy, Cg, t * Cpo), g[form], None, ws, 1,
                         accumulate=form in seen, math=math)
            seen.add(form)
        dw = _new(x, *weight.shape)
        check(lib().nvq_tconv_grad_combine(ptr(g["first"]), ptr(g["mid"]), ptr(g["last"]), Co, Ci, Cpi, ptr(dw), stream()),
              "nvq_tconv_grad_combine")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dx[t] = sum_k W_k^T dy[t + 1 - k]: a 1x1 conv over the dy frames t-1..t+1, taps in reverse order
            forms = {"first": (0, 2), "mid": (0, 3), "last": (1, 2)}
            packs = {k: K.conv_pack(_tconv_cat(weight, Cpo, k0, nk, True), False, nk * Cpo, math=math)
                     for k, (k0, nk) in forms.items() if k != "mid" or T > 2}
            for t in range(T):
                lo, hi = max(t - 1, 0), min(t + 1, T - 1)
                form = "first" if t == 0 else "last" if t == T - 1 else "mid"
                K.conv_forward(Sl(dy, (hi - lo + 1) * Cpo, lo * Cpo), packs[form], None, Sl(dx, Ci, t * Cpi), 1, cout_store=Cpi,
                               math=math)
        return dx, dw, None, None


class GroupMeanTC(torch.autograd.Function):
    """mean over the T frames of a time-in-channels tensor [B,H,W,T*Cp] -> fp32 [B,H,W,pad4(C)]
    (AdaptiveAvgPool3d((1,None,None)), frame_recovery.py:137,164-165)"""

    @staticmethod
    def forward(ctx, x, T: int, C: int):
        B, H, W, ld = x.shape
        Cp = ld // T
        y = _new(x, B, H, W, pad4(C))
        for t in range(T):
            cast_slice_(y, x, pad4(C), src_coff=t * Cp, alpha=1.0 / T, accumulate=t > 0)
        ctx.T, ctx.src = T, (x.dtype, ld, Cp)
        return y

    @staticmethod
    def backward(ctx, dy):
        T = ctx.T
        dtype, ld, Cp = ctx.src
        dy = dy.contiguous()
        B, H, W, c4 = dy.shape
        dx = _new(dy, B, H, W, ld, dtype=dtype, zero=Cp > c4)
        for t in range(T):
            cast_slice_(dx, dy, c4, dst_coff=t * Cp, alpha=1.0 / T)
        return dx, None, None


def bn_tc(x, mod, T: int, training: bool, relu: bool):
    """BatchNorm3d over a time-in-channels tensor: statistics over (B, T, H, W) = BatchNorm over the view [B, H, W*T, Cp]"""
    B, H, W, ld = x.shape
    return bn(x.view(B, H, W * T, ld // T), mod, training, relu).view(B, H, W, ld)


class ConvT(torch.autograd.Function):
    """nn.ConvTranspose2d(Ci, Co, 4, 2, 1, bias=False) (frame_recovery.py:283-304): phase-packed 3x3 conv to 4*Co channels
    + depth-to-space.  weight [Ci, Co, 4, 4]; [N,H,W,ld] -> [N,2H,2W,Co]."""

    @staticmethod
    def forward(ctx, x, weight, math: int):
        N, H, W, ld = x.shape
        Ci, Co = weight.shape[:2]
        assert Co % 4 == 0 and Ci <= ld
        w3 = _new(x, 4 * Co, Ci, 3, 3)
        check(lib().nvq_convt_pack(ptr(weight), Ci, Co, ptr(w3), stream()), "nvq_convt_pack")
        assert x.dtype == torch.float32 or Co % 8 == 0
        u = _new(x, N, H, W, 4 * Co, dtype=x.dtype)
        K.conv_forward(Sl(x), K.conv_pack(w3, False, ld, math=math), None, Sl(u), 3, math=math)
        y = _new(x, N, 2 * H, 2 * W, Co, dtype=x.dtype)
        check(lib().nvq_depth_space2(ptr(u), ptr(y), N, H, W, Co, 0, _bf(x), stream()), "nvq_depth_space2")
        ctx.save_for_backward(x, w3)
        ctx.math, ctx.wshape = math, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w3 = ctx.saved_tensors
        math = ctx.math
        N, H, W, ld = x.shape
        Ci, Co = ctx.wshape[:2]
        dy = dy.contiguous()
        du = _new(x, N, H, W, 4 * Co, dtype=dy.dtype)
        check(lib().nvq_depth_space2(ptr(dy), ptr(du), N, H, W, Co, 1, _bf(dy), stream()), "nvq_depth_space2")
        dw3 = torch.empty_like(w3)
        K.conv_wgrad(Sl(x), Ci, Sl(du), dw3, None, _ws(x), 3, math=math)
        dw = _new(x, *ctx.wshape)
        check(lib().nvq_convt_unpack_grad(ptr(dw3), Ci, Co, ptr(dw), stream()), "nvq_convt_unpack_grad")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            K.conv_forward(Sl(du), K.conv_pack(w3, True, 4 * Co, Ci, math=math), None, Sl(dx, Ci), 3, cout_store=ld, math=math)
        return dx, dw, None


class Stem7(torch.autograd.Function):
    """nn.Conv2d(4, Co, 7, 2, 3, bias=False) on a 4-channel NHWC image (frame_recovery.py:42-44); the image carries no gradient"""

    @staticmethod
    def forward(ctx, x4, weight, out_dtype=torch.float32):
        N, H, W, c = x4.shape
        Co = weight.shape[0]
        assert c == 4 and tuple(weight.shape[1:]) == (4, 7, 7) and Co % 8 == 0
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _new(x4, N, OH, OW, Co, dtype=out_dtype)
        check(lib().nvq_stem7_forward(ptr(x4), ptr(weight), N, H, W, Co, ptr(y), Co, _bf(y), stream()), "nvq_stem7_forward")
        ctx.save_for_backward(x4)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (x4,) = ctx.saved_tensors
        N, H, W, _ = x4.shape
        dy = dy.contiguous()
        dw = _new(x4, *ctx.wshape)
        ws = _ws(x4)
        check(lib().nvq_stem7_wgrad(ptr(x4), ptr(dy), dy.shape[-1], N, H, W, ctx.wshape[0], ptr(dw), ptr(ws), ws.numel() * 4,
                                    _bf(dy), stream()), "nvq_stem7_wgrad")
        return None, dw, None


# ----------------------------------------------------------------------------- BatchNorm (+ residual) (+ ReLU)
class BatchNorm(torch.autograd.Function):
    """nn.BatchNorm2d / BatchNorm3d over every pixel of x (train: batch statistics, running statistics updated in place;
    eval: running statistics), then (+ res), then ReLU when relu: covers BN, BN+ReLU (DepthwiseSeparableConv,
    TemporalConv3D, decoder) and ReLU(BN + identity) (ResidualBlock, efficient_layers.py:145-151)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, rmean, rvar, training: bool, relu: bool):
        N, H, W, ld = x.shape
        C = gamma.numel()
        npix = N * H * W
        assert res is None or res.dtype == x.dtype
        mean, invstd = _new(x, C), _new(x, C)
        ws = _ws(x)
        if training:
            check(lib().nvq_bn2_stats(ptr(x), ld, C, npix, BN_EPS, BN_MOMENTUM, ptr(mean), ptr(invstd), ptr(rmean), ptr(rvar),
                                      ptr(ws), ws.numel() * 4, _bf(x), stream()), "nvq_bn2_stats")
        else:
            check(lib().nvq_bn2_eval_stats(ptr(rmean), ptr(rvar), C, BN_EPS, ptr(mean), ptr(invstd), stream()), "nvq_bn2_eval_stats")
        y = torch.empty_like(x)
        check(lib().nvq_bn2_apply(ptr(x), ld, C, npix, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(res),
                                  res.shape[-1] if res is not None else 0, int(relu), ptr(y), ld, _bf(x), stream()), "nvq_bn2_apply")
        ctx.save_for_backward(x, gamma, beta, res, mean, invstd)
        ctx.training, ctx.relu = training, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, res, mean, invstd = ctx.saved_tensors
        N, H, W, ld = x.shape
        C = gamma.numel()
        dy = dy.contiguous()
        assert dy.dtype == x.dtype
        dx = torch.empty_like(x)
        dres = torch.empty_like(res) if res is not None else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws = _ws(x)
        check(lib().nvq_bn2_backward(ptr(dy), dy.shape[-1], ptr(x), ld, C, N * H * W, ptr(mean), ptr(invstd), ptr(gamma),
                                     ptr(beta), ptr(res), res.shape[-1] if res is not None else 0, int(ctx.relu),
                                     int(ctx.training), ptr(dx), ld, ptr(dres), dres.shape[-1] if dres is not None else 0,
                                     ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel() * 4, _bf(x), stream()), "nvq_bn2_backward")
        return dx, dgamma, dbeta, dres, None, None, None, None


# ----------------------------------------------------------------------------- pooling / resampling
class MaxPool(torch.autograd.Function):
    """nn.MaxPool2d(k, s, pad) / F.max_pool3d(x, (1,k,k)) on the image batch"""

    @staticmethod
    def forward(ctx, x, k: int, s: int, pad: int):
        N, H, W, ld = x.shape
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        y = _new(x, N, OH, OW, ld, dtype=x.dtype)
        idx = _new(x, N, OH, OW, ld, dtype=torch.uint8)
        check(lib().nvq_maxpool_forward(ptr(x), ld, N, H, W, k, s, pad, ptr(y), ptr(idx), _bf(x), stream()), "nvq_maxpool_forward")
        ctx.save_for_backward(idx)
        ctx.args = (N, H, W, ld, k, s, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, ld, k, s, pad = ctx.args
        dy = dy.contiguous()
        dx = _new(dy, N, H, W, ld, dtype=dy.dtype)
        check(lib().nvq_maxpool_backward(ptr(dy), ptr(idx), ld, N, H, W, k, s, pad, ptr(dx), _bf(dy), stream()), "nvq_maxpool_backward")
        return dx, None, None, None


class Subsample2(torch.autograd.Function):
    """x[:, ::2, ::2]: the input side of nn.Conv2d(kernel 1, stride 2) (frame_recovery.py:71)"""

    @staticmethod
    def forward(ctx, x):
        N, H, W, ld = x.shape
        y = _new(x, N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, ld, dtype=x.dtype)
        check(lib().nvq_subsample2(ptr(x), ld, N, H, W, ptr(y), 0, _bf(x), stream()), "nvq_subsample2")
        ctx.shape = (N, H, W, ld)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, ld = ctx.shape
        dy = dy.contiguous()
        dx = _new(dy, N, H, W, ld, dtype=dy.dtype)
        check(lib().nvq_subsample2(ptr(dy), ld, N, H, W, ptr(dx), 1, _bf(dy), stream()), "nvq_subsample2")
        return dx


class Resize(torch.autograd.Function):
    """F.interpolate(size=(OH, OW), mode='bilinear', align_corners=False)"""

    @staticmethod
    def forward(ctx, x, OH: int, OW: int):
        N, H, W, ld = x.shape
        assert x.dtype == torch.float32
        y = _new(x, N, OH, OW, ld)
        check(lib().nvq_bilinear_resize(ptr(x), ld, N, H, W, OH, OW, ptr(y), 0, stream()), "nvq_bilinear_resize")
        ctx.args = (N, H, W, ld, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, ld, OH, OW = ctx.args
        dy = dy.contiguous()
        dx = _new(dy, N, H, W, ld)
        check(lib().nvq_bilinear_resize(ptr(dy), ld, N, H, W, OH, OW, ptr(dx), 1, stream()), "nvq_bilinear_resize")
        return dx, None, None


class DepthToSpace2(torch.autograd.Function):
    """[N,H,W,4*Co] (phase-major) -> [N,2H,2W,Co]; with Co = out_channels and the phases ordered (i*2+j) this is NOT
    nn.PixelShuffle's channel order (c*4 + i*2 + j) - see PixelShuffle2 below for that one."""

    @staticmethod
    def forward(ctx, u):
        N, H, W, c = u.shape
        Co = c // 4
        y = _new(u, N, 2 * H, 2 * W, Co, dtype=u.dtype)
        check(lib().nvq_depth_space2(ptr(u), ptr(y), N, H, W, Co, 0, _bf(u), stream()), "nvq_depth_space2")
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H2, W2, Co = dy.shape
        du = _new(dy, N, H2 // 2, W2 // 2, 4 * Co, dtype=dy.dtype)
        check(lib().nvq_depth_space2(ptr(dy.contiguous()), ptr(du), N, H2 // 2, W2 // 2, Co, 1, _bf(dy), stream()), "nvq_depth_space2")
        return du


class GroupMean(torch.autograd.Function):
    """mean over the T leading image groups of a time-major batch [T*B, ...] -> [B, ...] (AdaptiveAvgPool3d((1,None,None)),
    frame_recovery.py:137,164-165)"""

    @staticmethod
    def forward(ctx, x, T: int, C: int):
        NB, H, W, ld = x.shape
        B = NB // T
        y = _new(x, B, H, W, pad4(C))                        # the mean is fp32 whatever the storage type of x
        for t in range(T):
            cast_slice_(y, x[t * B:(t + 1) * B], pad4(C), alpha=1.0 / T, accumulate=t > 0)
        ctx.T, ctx.src = T, (x.dtype, ld)
        return y

    @staticmethod
    def backward(ctx, dy):
        T = ctx.T
        dtype, ld = ctx.src
        dy = dy.contiguous()
        B, H, W, c4 = dy.shape
        dx = _new(dy, T * B, H, W, ld, dtype=dtype, zero=ld > c4)
        for t in range(T):
            cast_slice_(dx[t * B:(t + 1) * B], dy, c4, alpha=1.0 / T)
        return dx, None, None


class Cat2(torch.autograd.Function):
    """torch.cat([a, b], dim=channels) for two tensors whose channel counts are multiples of 4"""

    @staticmethod
    def forward(ctx, a, b):
        N, H, W, ca = a.shape
        cb = b.shape[-1]
        y = _new(a, N, H, W, ca + cb)
        K.axpy_slice(Sl(y, ca, 0), Sl(a), 1.0, accumulate=False)
        K.axpy_slice(Sl(y, cb, ca), Sl(b), 1.0, accumulate=False)
        ctx.c = (ca, cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        ca, cb = ctx.c
        dy = dy.contiguous()
        N, H, W, _ = dy.shape
        da, db = _new(dy, N, H, W, ca), _new(dy, N, H, W, cb)
        K.axpy_slice(Sl(da), Sl(dy, ca, 0), 1.0, accumulate=False)
        K.axpy_slice(Sl(db), Sl(dy, cb, ca), 1.0, accumulate=False)
        return da, db


# ----------------------------------------------------------------------------- attention / fusion / tail
class CBAMFn(torch.autograd.Function):
    """CBAM (efficient_layers.py:154-228) on the kernels the SR aggregator uses: GAP -> MLP -> sigmoid -> channel scale,
    channel mean/max -> 7x7 conv -> sigmoid -> spatial scale.  C a power of two in [16, 256]."""

    @staticmethod
    def forward(ctx, x, w1, w2, w7):
        N, H, W, C = x.shape
        R = w1.shape[0]
        nblk = int(lib().nvq_gap_blocks(H, W))
        part = _new(x, N, nblk, C)
        check(lib().nvq_gap_partial(ptr(x), C, C, N, H, W, ptr(part), stream()), "nvq_gap_partial")
        gap, hid, ca = _new(x, N, C), _new(x, N, R), _new(x, N, C)
        K.cbam_channel(part, nblk, C, R, N, H * W, w1, w2, gap, hid, ca)
        sm, amax, sa = _new(x, N, H, W, 2), _new(x, N, H, W, dtype=torch.int32), _new(x, N, H, W)
        K.cbam_pool(x, ca, sm, amax)
        y = torch.empty_like(x)
        K.cbam_spatial_apply(x, ca, sm, w7, sa, Sl(y))
        ctx.save_for_backward(x, w1, w2, w7, gap, hid, ca, sm, amax, sa)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, w7, gap, hid, ca, sm, amax, sa = ctx.saved_tensors
        N, H, W, C = x.shape
        R = w1.shape[0]
        dy = dy.contiguous()
        ws = _ws(x)
        dpre = _new(x, N, H, W)
        K.cbam_bwd_spatial_pre(Sl(dy), x, ca, sa, dpre)
        dsm, dw7 = _new(x, N, H, W, 2), torch.empty_like(w7)
        K.cbam_bwd_spatial_conv(dpre, sm, w7, dsm, dw7, ws)
        nblk = K.tsum_blocks(H, W)
        dx, dca_partial = torch.empty_like(x), _new(x, N, nblk, C)
        K.cbam_bwd_scale(Sl(dy), x, ca, sa, dsm, amax, dx, dca_partial)
        dw1, dw2, dgap_pix = torch.empty_like(w1), torch.empty_like(w2), _new(x, N, C)
        K.cbam_bwd_channel(dca_partial, nblk, C, R, N, H * W, w1, w2, gap, hid, ca, dw1, dw2, dgap_pix)
        check(lib().nvq_add_image_channel(ptr(dx), C, C, N, H, W, ptr(dgap_pix), stream()), "nvq_add_image_channel")
        return dx, dw1, dw2, dw7


class FusionMix(torch.autograd.Function):
    """aligned + softmax(logits)[0] * mean_c(spatial) + softmax(logits)[1] * mean_c(temporal) (frame_recovery.py:239-254)"""

    @staticmethod
    def forward(ctx, aligned, logits, sp, tp):
        N, H, W, C = aligned.shape
        npix = N * H * W
        y = torch.empty_like(aligned)
        attn, means = _new(aligned, npix, 2), _new(aligned, npix, 2)
        check(lib().nvq_fusion_mix_forward(ptr(aligned), ptr(logits), logits.shape[-1], ptr(sp), sp.shape[-1], ptr(tp),
                                           tp.shape[-1], C, npix, ptr(y), ptr(attn), ptr(means), stream()), "nvq_fusion_mix_forward")
        ctx.save_for_backward(attn, means)
        ctx.shapes = (aligned.shape, logits.shape, sp.shape, tp.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        attn, means = ctx.saved_tensors
        sa, sl, ss, st = ctx.shapes
        dy = dy.contiguous()
        dl, dsp, dtp = _new(dy, *sl), _new(dy, *ss), _new(dy, *st)
        check(lib().nvq_fusion_mix_backward(ptr(dy), ptr(attn), ptr(means), sa[-1], sa[0] * sa[1] * sa[2], ptr(dl), sl[-1],
                                            ptr(dsp), ss[-1], ptr(dtp), st[-1], stream()), "nvq_fusion_mix_backward")
        return dy, dl, dsp, dtp


class Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        check(lib().nvq_tanh(ptr(x), None, x.numel(), ptr(y), 0, stream()), "nvq_tanh")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        check(lib().nvq_tanh(ptr(dy.contiguous()), ptr(y), y.numel(), ptr(dx), 1, stream()), "nvq_tanh")
        return dx


class MaskBlend(torch.autograd.Function):
    """frame * (1 - mask) + rec * mask (frame_recovery.py:439-440): frame (N,C,H,W) and mask (N,1,H,W) are data, rec is
    [N,H,W,ld]; returns (N,C,H,W)."""

    @staticmethod
    def forward(ctx, frame, rec, mask):
        N, C, H, W = frame.shape
        out = torch.empty_like(frame)
        check(lib().nvq_mask_blend(ptr(frame), ptr(rec), rec.shape[-1], ptr(mask), N, C, H, W, ptr(out), stream()), "nvq_mask_blend")
        ctx.save_for_backward(mask)
        ctx.args = (N, C, H, W, rec.shape[-1])
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        N, C, H, W, ld = ctx.args
        drec = _new(dout, N, H, W, ld)
        check(lib().nvq_mask_blend_backward(ptr(dout.contiguous()), ptr(mask), N, C, H, W, ptr(drec), ld, stream()),
              "nvq_mask_blend_backward")
        return None, drec, None


class PixelShuffleNCHW(torch.autograd.Function):
    """nn.PixelShuffle(s) of an NHWC conv output u [N,H,W,ld] (channel c*s*s + i*s + j) into (N,C,H*s,W*s)
    (efficient_layers.py:101-106): the shuffle kernel of the SR tail with strength 0 for the bicubic skip is not what is
    wanted here, so the plain gather / scatter pair of upsample.hip is used through its 'no clamp' form."""

    @staticmethod
    def forward(ctx, u, C: int, s: int):
        N, H, W, ld = u.shape
        out = _new(u, N, C, H * s, W * s)
        check(lib().nvq_pixel_shuffle(ptr(u), ld, N, C, H, W, s, ptr(out), 0, stream()), "nvq_pixel_shuffle")
        ctx.args = (N, C, H, W, s, ld)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, C, H, W, s, ld = ctx.args
        du = _new(dout, N, H, W, ld)
        check(lib().nvq_pixel_shuffle(ptr(du), ld, N, C, H, W, s, ptr(dout.contiguous()), 1, stream()), "nvq_pixel_shuffle")
        return du, None, None


class Correlation(torch.autograd.Function):
    """LiteFlowNetCorrelation (efficient_layers.py:313-343, d = 4): [N,H,W,ld] x 2 -> [N,H,W,96] (81 channels used),
    exact-fp32 kernels of the SR motion estimator"""

    @staticmethod
    def forward(ctx, x1, x2, C: int):
        N, H, W, _ = x1.shape
        out = _new(x1, N, H, W, _engine.CORR_LD)
        K.correlation_forward(Sl(x1, C), Sl(x2, C), out)
        ctx.save_for_backward(x1, x2)
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dcorr):
        x1, x2 = ctx.saved_tensors
        C = ctx.C
        dcorr = dcorr.contiguous()
        dx1, dx2 = torch.zeros_like(x1), torch.zeros_like(x2)
        K.correlation_backward(1, dcorr, Sl(x2, C), Sl(dx1, C), True)
        K.correlation_backward(2, dcorr, Sl(x1, C), Sl(dx2, C), True)
        return dx1, dx2, None


class ScaleAdd(torch.autograd.Function):
    """alpha * a + b over the first C channels of two [N,H,W,ld] tensors (the `features = body(h) + h` skip of the feature
    extractor, super_resolution.py:53, and the 0.2-scaled residual of a dense block, :253)"""

    @staticmethod
    def forward(ctx, a, b, alpha: float):
        assert a.shape == b.shape
        y = b.clone()
        K.axpy_slice(Sl(y), Sl(a), float(alpha), accumulate=True)
        ctx.alpha = float(alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = torch.empty_like(dy)
        K.axpy_slice(Sl(da), Sl(dy), ctx.alpha, accumulate=False)
        return da, dy, None


class SoftmaxWeightedSum(torch.autograd.Function):
    """TemporalAggregator's  softmax over the T attention logits + sum_t aligned_t * attn_t  (super_resolution.py:174,
    203-204).  aligned [N,H,W,T*C] frame-major, logits [N,H,W,>=T] -> weighted [N,H,W,C]."""

    @staticmethod
    def forward(ctx, aligned, logits, T: int, C: int):
        N, H, W, _ = aligned.shape
        attn, weighted = _new(aligned, N, H, W, K.pad4(T)), _new(aligned, N, H, W, C)
        gap_partial = _new(aligned, N, K.tsum_blocks(H, W), C)
        K.tsum_forward(aligned, logits, T, C, attn, weighted, gap_partial)
        ctx.save_for_backward(aligned, attn)
        ctx.T, ctx.C, ctx.lp = T, C, logits.shape[-1]
        return weighted

    @staticmethod
    def backward(ctx, dweighted):
        aligned, attn = ctx.saved_tensors
        N, H, W, _ = aligned.shape
        dweighted = dweighted.contiguous()
        daligned, dlogits = torch.empty_like(aligned), _new(aligned, N, H, W, ctx.lp, zero=True)
        K.tsum_backward(dweighted, _new(aligned, N, ctx.C, zero=True), aligned, attn, ctx.T, ctx.C, daligned, dlogits)
        return daligned, dlogits, None, None


class Warp(torch.autograd.Function):
    """warp_features (super_resolution.py:104-143): bilinear sampling of `feat` at (x + flow_x, y + flow_y), zero padding,
    align_corners=True; gradients to the features and to the flow.  feat [N,H,W,ld] (C channels), flow [N,H,W,4]."""

    @staticmethod
    def forward(ctx, feat, flow, C: int):
        out = _new(feat, *feat.shape, zero=feat.shape[-1] > C)
        K.warp_forward(Sl(feat, C), flow, Sl(out, C))
        ctx.save_for_backward(feat, flow)
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, flow = ctx.saved_tensors
        dout = dout.contiguous()
        dfeat, dflow = torch.empty_like(feat), torch.empty_like(flow)
        K.warp_backward(Sl(dout, ctx.C), Sl(feat, ctx.C), flow, Sl(dfeat, ctx.C), dflow, overwrite=True)
        return dfeat, dflow, None


# ----------------------------------------------------------------------------- functional helpers used by the modules
def bn(x, mod, training: bool, relu: bool, res=None):
    """apply an nn.BatchNorm2d / BatchNorm3d holder `mod` (its buffers are updated in place in training mode)"""
    if training:
        mod.num_batches_tracked.add_(1)
    return BatchNorm.apply(x, mod.weight, mod.bias, res, mod.running_mean, mod.running_var, training, relu)
