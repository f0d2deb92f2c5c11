"""Kernel schedule of SuperResolutionNet forward and backward on libnvq.

The op order follows the reference forward (super_resolution.py:327-391); every arithmetic
step is a libnvq kernel launch on the current HIP stream, PyTorch only owns the buffers.
Internal activations are fp32 NHWC.  Frames are processed in "slot" order
[centre, other frames...] so the T feature-extractor calls of the reference become one
batched launch per layer (BatchNorm statistics stay per frame, see nvq_bn_stats).

Dense-block concatenation (torch.cat at super_resolution.py:249,252) never happens: each
block owns one [B,H,W,F+160] buffer, layer i reads channels [0, F+32i) and writes
[F+32i, F+32i+32); the backward accumulates into a gradient buffer of the same layout.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import _nvq as K
from ._nvq import Sl

GROWTH = 32   # ResidualDenseBlock growth_rate (fixed, super_resolution.py:227)
LAYERS = 5    # ResidualDenseBlock num_layers   (fixed, super_resolution.py:228)
CORR_LD = 96  # 81 correlation channels stored in a 96-channel NHWC buffer (pad = 0)
BN_EPS = 1e-5
BN_MOM = 0.1

_ws_cache: Dict[torch.device, torch.Tensor] = {}

# When set to a dict, backward() stores intermediate gradients in it (tests / debugging only).
DEBUG_CAPTURE: Optional[dict] = None

_TAIL_ROWS = int(os.environ.get("NVQ_TAIL_ROWS", "0"))   # 4: the four-wave rdb_tail kernel (A/B switch, same results)


def _fused_tail_ok(math: int, x: torch.Tensor, Cimg: int, scale: int) -> bool:
    """nvq_upsampler_tail_forward's domain: bf16 MFMA mode, a bf16-stored input, scale 2..4 with Cimg * scale^2 <= 16 / 32 / 64.
    (The exact-fp32 mode keeps the conv + nvq_shuffle_bicubic_clamp pair; NVQ_FUSED_TAIL=0 selects it in bf16 mode too.)"""
    return (math == K.MATH_BF16 and x.dtype == torch.bfloat16 and scale in (2, 3, 4)
            and Cimg * scale * scale <= {2: 16, 3: 32, 4: 64}[scale] and os.environ.get("NVQ_FUSED_TAIL", "1") != "0")


def _capture(name: str, t) -> None:
    """t: a tensor, or a callable producing one (evaluated only while a capture is requested)"""
    if DEBUG_CAPTURE is not None:
        if callable(t):
            t = t()
        DEBUG_CAPTURE[name] = t.clone() if torch.is_tensor(t) else t


class _Packs:
    """The packed weights of one pass.  `prefetch` packs a list of (name, transpose, cin_store, keep) in one launch per 48 jobs
    (K.conv_pack_many); `get` returns such a pack, or packs a single weight the list did not name."""

    def __init__(self, P: Dict[str, torch.Tensor], math: int):
        self.P, self.math, self.done = P, math, {}

    def prefetch(self, reqs) -> None:
        reqs = [r for r in dict.fromkeys(reqs) if r not in self.done]
        outs = K.conv_pack_many([(self.P[n], tr, cs, keep) for n, tr, cs, keep in reqs], self.math)
        self.done.update(zip(reqs, outs))

    def get(self, name: str, transpose: bool, cin_store: int, keep: Optional[int] = None) -> torch.Tensor:
        key = (name, transpose, cin_store, keep)
        wp = self.done.get(key)
        if wp is None:
            wp = self.done[key] = K.conv_pack(self.P[name], transpose, cin_store, keep, math=self.math)
        return wp


def workspace(device) -> torch.Tensor:
    ws = _ws_cache.get(device)
    if ws is None:
        ws = torch.empty(K.wgrad_workspace_bytes() // 4 + (4 << 20), dtype=torch.float32, device=device)
        _ws_cache[device] = ws
    return ws


_ws_multi_cache: Dict = {}
SMALL_STEP_PIXELS = int(os.environ.get("NVQ_SMALL_STEP_PIXELS", str(16 * 128 * 128)))   # batch x H x W up to which a step is launch-bound


def workspace_slots(device, n: int) -> "list[torch.Tensor]":
    """n weight-gradient workspaces that can be in use at the same time (deferred reduces of a dense block: small steps only)"""
    ws = _ws_multi_cache.get((device, n))
    if ws is None:
        k = K.wgrad_workspace_bytes() // 4 + 1024
        flat = torch.empty(n * k, dtype=torch.float32, device=device)
        ws = _ws_multi_cache[(device, n)] = [flat[i * k:(i + 1) * k] for i in range(n)]
    return ws


class _ReduceQueue:
    """Deferred weight-gradient reduces of a launch-bound step: every _wgrad takes the next free workspace slot and leaves its
    reduce in the queue; a full queue (and the end of the pass) is one nvq_wgrad_reduce_batch launch."""
    SLOTS = 8

    def __init__(self, device):
        self.slots = workspace_slots(device, self.SLOTS)
        self.jobs: list = []

    def ws(self) -> torch.Tensor:
        if len(self.jobs) == self.SLOTS:
            self.flush()
        return self.slots[len(self.jobs)]

    def flush(self) -> None:
        K.wgrad_reduce_batch(self.jobs)


class Geometry:
    def __init__(self, frames: torch.Tensor, F: int, nblocks: int, scale: int):
        self.B, self.T, self.Cimg, self.H, self.W = frames.shape
        self.F, self.NB, self.s = F, nblocks, scale
        self.c = self.T // 2
        self.slots = [self.c] + [t for t in range(self.T) if t != self.c]
        self.NI = self.T * self.B
        self.NO = (self.T - 1) * self.B
        self.CAT = F + GROWTH * LAYERS
        # Pixel stride of the dense-block buffers: a multiple of 64 channels, so that every pixel of the bf16 buffer
        # starts on a 128-B line (F=64: 224 -> 256).  The 32-channel (64 B) pieces the conv kernels fetch per pixel
        # and chunk then never share a line with the neighbouring pixel: measured 171 -> 146 us on the cin=192 conv.
        self.CATLD = (self.CAT + 63) // 64 * 64
        self.Tp = K.pad4(self.T)
        self.U = self.Cimg * scale * scale
        self.Up = K.pad4(self.U)
        self.R = F // 16


def _new(dev, *shape, dtype=torch.float32, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=dev)


class Saved:
    """Everything the backward needs; also the carrier of return_intermediate tensors."""
    pass


def _extract(P: Dict[str, torch.Tensor], frames: torch.Tensor, slots, F: int, training: bool, math: int,
             act_dtype: torch.dtype, outA: Sl, split_images: int, outB: Optional[Sl], sv) -> None:
    """FeatureExtractor.forward (super_resolution.py:40-53) for the T = len(slots) frames of every clip, batched in slot
    order; the features of the first `split_images` images go to outA, the rest to outB.  Saves what backward needs in sv."""
    B, _, _, H, W = frames.shape
    T = len(slots)
    NI = T * B
    dev = frames.device
    ws = workspace(dev)
    feat0 = _new(dev, NI, H, W, F, dtype=act_dtype)
    # bf16 mode: the head's weight gradient runs on the matrix cores (conv_wgrad over the frames as bf16 NHWC-8)
    sv.img8 = _new(dev, NI, H, W, 8, dtype=torch.bfloat16) if (training and K.dwconv_bn_fusable(feat0, F)) else None
    K.head_forward(frames, slots, P["feature_extractor.head.0.weight"], P["feature_extractor.head.0.bias"], feat0,
                   img8=sv.img8, math=math)
    sv.feat0 = feat0
    sv.dws, sv.pws, sv.acts, sv.bn_mean, sv.bn_invstd = [], [], [], [], []
    cur, cur_bn = feat0, None         # cur_bn: BatchNorm + ReLU still to be applied to `cur` (fused into the consumer)
    sv.dw_in = []                     # (tensor, bn) the depthwise conv of layer k was fed with
    fuse_bn = K.dwconv_bn_fusable(feat0, F)
    fused_fwd = fuse_bn and F == 64 and math == K.MATH_BF16 and os.environ.get("NVQ_FUSED_DWPW", "1") != "0"
    for k in range(3):
        pre = f"feature_extractor.body.{k}."
        d = _new(dev, NI, H, W, F, dtype=act_dtype)
        p = _new(dev, NI, H, W, F, dtype=act_dtype)
        mean, invstd = _new(dev, T, F), _new(dev, T, F)
        order = [slots.index(t) for t in range(T)]
        if fused_fwd:
            # depthwise -> pointwise -> BatchNorm sums in one pass over the tensors (nvq_dwpw_forward)
            K.dwpw_forward(cur, cur_bn, P[pre + "depthwise.weight"], P[pre + "pointwise.weight"], d, p, B,
                           order if training else None, mean, invstd, P[pre + "bn.running_mean"],
                           P[pre + "bn.running_var"], ws, BN_EPS, BN_MOM)
        else:
            K.dwconv_forward(cur, P[pre + "depthwise.weight"], d, bn=cur_bn)
            wp = K.conv_pack(P[pre + "pointwise.weight"], False, F, math=math)
            K.conv_forward(Sl(d), wp, None, Sl(p), 1, math=math)
            if training:
                K.bn_stats(p, B, order, mean, invstd, P[pre + "bn.running_mean"], P[pre + "bn.running_var"], ws, BN_EPS,
                           BN_MOM)
        sv.dw_in.append((cur, cur_bn))
        if training:
            P[pre + "bn.num_batches_tracked"].add_(T)
        else:
            K.bn_eval_stats(P[pre + "bn.running_mean"], P[pre + "bn.running_var"], T, mean, invstd, BN_EPS)
        if k < 2 and fuse_bn:
            # relu(bn(p)) is evaluated by the next depthwise conv (and by its weight gradient) while it stages p
            r, cur_bn = p, (mean, invstd, P[pre + "bn.weight"], P[pre + "bn.bias"], B)
        elif k < 2:
            r, cur_bn = _new(dev, NI, H, W, F, dtype=act_dtype), None
            K.bn_apply_relu(p, B, mean, invstd, P[pre + "bn.weight"], P[pre + "bn.bias"], None, Sl(r), NI)
        else:
            # features = relu(bn(.)) + head features
            r = None
            K.bn_apply_relu(p, B, mean, invstd, P[pre + "bn.weight"], P[pre + "bn.bias"], feat0, outA, split_images, outB)
        sv.dws.append(d); sv.pws.append(p); sv.acts.append(r); sv.bn_mean.append(mean); sv.bn_invstd.append(invstd)
        cur = r


def forward(P: Dict[str, torch.Tensor], frames: torch.Tensor, F: int, nblocks: int, scale: int,
            training: bool, math: int = K.MATH_F32, act_dtype: torch.dtype = torch.float32,
            features: Optional[torch.Tensor] = None) -> "tuple[torch.Tensor, Saved]":
    """act_dtype: storage type of the conv-internal tensors (dense-block concat buffers, flow-net and attention
    hidden activations and, in backward, their gradients).  torch.bfloat16 needs math == MATH_BF16; every
    tensor a non-conv kernel touches stays fp32."""
    assert act_dtype == torch.float32 or math == K.MATH_BF16
    g = Geometry(frames, F, nblocks, scale)
    dev = frames.device
    B, T, H, W, NI, NO, c = g.B, g.T, g.H, g.W, g.NI, g.NO, g.c
    ws = workspace(dev)
    sv = Saved()
    sv.g, sv.frames, sv.training, sv.math, sv.act_dtype = g, frames, training, math, act_dtype
    # every forward weight pack of the step in two launches (57 single-pack launches otherwise)
    packs = _Packs(P, math)
    reqs = []
    if NO:
        reqs += [(f"motion_estimator.flow_net.{idx}.weight", False, cs, None)
                 for idx, cs in zip((0, 2, 4, 6), (CORR_LD, K.pad4(128), K.pad4(64), K.pad4(32)))]
    reqs += [("temporal_aggregator.attention.0.weight", False, T * F, None), ("temporal_aggregator.attention.2.weight", False, F, None),
             ("temporal_aggregator.attention.4.weight", False, F, None)]
    for k in range(nblocks):
        reqs += [(f"residual_blocks.{k}.layers.{i}.0.weight", False, F + GROWTH * i, None) for i in range(LAYERS)]
        reqs.append((f"residual_blocks.{k}.lff.weight", False, g.CAT, None))
    reqs += [("gff.0.weight", False, F, None), ("upsampler.conv.weight", False, F, None)]
    packs.prefetch(reqs)

    # ---- feature extractor, all T frames in one batch (slot order)
    # bf16 activation mode: the frames' features (`aligned`: centre frame + warped neighbours, `feat_oth`: the neighbours
    # before warping) are stored as bf16 like every other conv-internal tensor; their non-conv readers (correlation on
    # the matrix cores, warp, softmax-weighted sum) take a storage flag.  NVQ_BF16_FEATURES=0 keeps them fp32.
    feat_dtype = act_dtype if (act_dtype == torch.bfloat16 and math == K.MATH_BF16 and F in (32, 64)
                               and os.environ.get("NVQ_BF16_FEATURES", "1") != "0") else torch.float32
    aligned = _new(dev, B, H, W, T * F, dtype=feat_dtype)
    feat_oth = _new(dev, max(NO, 1), H, W, F, dtype=feat_dtype)
    sv.aligned, sv.feat_oth = aligned, feat_oth
    if features is None:
        _extract(P, frames, g.slots, F, training, math, act_dtype, Sl(aligned, F, c * F), B, Sl(feat_oth) if NO else None, sv)
    else:
        # inference with cached per-frame features (extract_features), [T,B,H,W,F] in time order: no extractor launches
        assert not training and tuple(features.shape) == (T, B, H, W, F) and features.dtype == torch.float32
        sv.feat0 = None                                      # marks the state as not differentiable
        aligned[..., c * F:(c + 1) * F].copy_(features[c])
        for j in range(1, T):
            feat_oth[(j - 1) * B:j * B].copy_(features[g.slots[j]])
    center = Sl(aligned, F, c * F)

    # ---- motion: correlation -> flow net -> warp, the T-1 reference frames batched
    if NO:
        # bf16 mode: correlation on the matrix cores, stored as bf16 with the same 96-channel pixel stride (192 B: the 16 pixels a
        # workgroup writes per row are 24 whole 128-B lines, and the flow net's three 32-channel chunks are 64-B pieces that never
        # straddle a line; a 128-channel stride moved a third more bytes through six passes for the same step time)
        corr_bf16 = act_dtype == torch.bfloat16 and F in (32, 64)
        corr = _new(dev, NO, H, W, CORR_LD, dtype=act_dtype if corr_bf16 else torch.float32)
        K.correlation_forward(Sl(feat_oth), center, corr, math=math)
        chans = [81, 128, 64, 32, 2]
        x = Sl(corr, CORR_LD, 0)
        sv.flow_acts = [corr]
        # bf16 mode, training: the hidden layers also leave their ReLU masks as one bit per channel (the input-gradient conv of the
        # layer behind reads 4 bytes per 32 channels instead of the 64-byte activations, as in the dense blocks)
        act_bits = (training and act_dtype == torch.bfloat16 and math == K.MATH_BF16
                    and os.environ.get("NVQ_RELU_BITS", "1") != "0")
        sv.flow_bits = [None] * 5
        for li, idx in enumerate((0, 2, 4, 6)):
            w = P[f"motion_estimator.flow_net.{idx}.weight"]
            wp = packs.get(f"motion_estimator.flow_net.{idx}.weight", False, x.c)
            last = idx == 6
            y = _new(dev, NO, H, W, K.pad4(chans[li + 1]), dtype=torch.float32 if last else act_dtype)
            # (more than 32 output channels: only the channel-split kernel of a bf16 input writes bits)
            bits = (_new(dev, NO, H, W, chans[li + 1] // 32, dtype=torch.int32)
                    if (act_bits and not last and (x.bf16 or chans[li + 1] <= 32)) else None)
            K.conv_forward(x, wp, P[f"motion_estimator.flow_net.{idx}.bias"], Sl(y, chans[li + 1]), 3,
                           relu=not last, cout_store=K.pad4(chans[li + 1]), math=math, bits=bits, bits_mode=1 if bits is not None else 0)
            sv.flow_acts.append(y)
            sv.flow_bits[li + 1] = bits
            x = Sl(y)
        flow = sv.flow_acts[-1]
        for j in range(1, T):
            t = g.slots[j]
            K.warp_forward(Sl(feat_oth).images((j - 1) * B, j * B), flow[(j - 1) * B:j * B], Sl(aligned, F, t * F))
    sv.flow = sv.flow_acts[-1] if NO else None

    # ---- temporal aggregation
    a1, a2 = _new(dev, B, H, W, F, dtype=act_dtype), _new(dev, B, H, W, F, dtype=act_dtype)
    logits = _new(dev, B, H, W, g.Tp)
    sv.a1_bits = (_new(dev, B, H, W, F // 32, dtype=torch.int32)
                  if (training and act_dtype == torch.bfloat16 and math == K.MATH_BF16 and F % 32 == 0
                      and (aligned.dtype == torch.bfloat16 or F <= 32) and os.environ.get("NVQ_RELU_BITS", "1") != "0") else None)
    K.conv_forward(Sl(aligned), packs.get("temporal_aggregator.attention.0.weight", False, T * F),
                   P["temporal_aggregator.attention.0.bias"], Sl(a1), 3, relu=True, math=math, bits=sv.a1_bits,
                   bits_mode=1 if sv.a1_bits is not None else 0)
    K.conv_forward(Sl(a1), packs.get("temporal_aggregator.attention.2.weight", False, F),
                   P["temporal_aggregator.attention.2.bias"], Sl(a2), 3, relu=True, math=math)
    K.conv_forward(Sl(a2), packs.get("temporal_aggregator.attention.4.weight", False, F),
                   P["temporal_aggregator.attention.4.bias"], Sl(logits, T), 3, cout_store=g.Tp, math=math)
    nblk = K.tsum_blocks(H, W)
    # bf16 activation mode: the aggregated features (the CBAM's input; one write, four reads per step) and the two gradients around
    # the CBAM backward are stored as bf16 like the tensors on either side of them; the pooled sums (GAP, channel mean / max,
    # dca) are formed from the unrounded values inside the kernels.  NVQ_BF16_CBAM=0 keeps them fp32.
    cbam16 = (act_dtype == torch.bfloat16 and math == K.MATH_BF16 and nblocks > 0
              and os.environ.get("NVQ_BF16_CBAM", "1") != "0")
    sv.cbam_dtype = torch.bfloat16 if cbam16 else torch.float32
    attn, weighted = _new(dev, B, H, W, g.Tp), _new(dev, B, H, W, F, dtype=sv.cbam_dtype)
    gap_partial = _new(dev, B, nblk, F)
    K.tsum_forward(aligned, logits, T, F, attn, weighted, gap_partial)
    gap, hid, ca = _new(dev, B, F), _new(dev, B, g.R), _new(dev, B, F)
    w1 = P["temporal_aggregator.refine.channel_attention.fc.0.weight"]
    w2 = P["temporal_aggregator.refine.channel_attention.fc.2.weight"]
    w7 = P["temporal_aggregator.refine.spatial_attention.conv.weight"]
    K.cbam_channel(gap_partial, nblk, F, g.R, B, H * W, w1, w2, gap, hid, ca)
    sm, amax, sa = _new(dev, B, H, W, 2), _new(dev, B, H, W, dtype=torch.int32), _new(dev, B, H, W)
    K.cbam_pool(weighted, ca, sm, amax)
    # bf16 mode: slice-planar dense-block buffers (K.CatBuf: x and every growth slice compact tensors of their own, whole
    # 128-B lines written by every layer); NVQ_PLANAR=0 or fp32 storage: one interleaved [B,H,W,CATLD] tensor per block
    planar = (act_dtype == torch.bfloat16 and math == K.MATH_BF16 and F in (32, 64, 128)
              and os.environ.get("NVQ_PLANAR", "1") != "0")
    cats = [K.CatBuf(dev, B, H, W, F, LAYERS, g.CATLD, act_dtype, planar) for _ in range(nblocks)]
    sv.planar = planar
    # (with no dense blocks the CBAM kernel, an fp32 writer, fills this tensor)
    resout = _new(dev, B, H, W, F, dtype=act_dtype if nblocks else torch.float32)

    def xloc(k):  # where the input of block k / the output of block k-1 lives
        return cats[k].x() if k < nblocks else Sl(resout)
    K.cbam_spatial_apply(weighted, ca, sm, w7, sa, xloc(0))
    sv.a1, sv.a2, sv.attn, sv.weighted = a1, a2, attn, weighted
    sv.gap, sv.hid, sv.ca, sv.sm, sv.amax, sv.sa, sv.nblk = gap, hid, ca, sm, amax, sa, nblk
    sv.cats, sv.resout = cats, resout

    # ---- residual dense blocks
    K.TIMER_TAG = "rdb"
    # bf16 mode: the dense layers also emit their ReLU masks as one bit per channel (4 B per pixel) for the backward
    use_bits = training and act_dtype == torch.bfloat16
    sv.bits = [[_new(dev, B, H, W, dtype=torch.int32) for _ in range(LAYERS)] for _ in range(nblocks)] if use_bits else None
    fuse_tail = (act_dtype == torch.bfloat16 and F == 64        # last dense layer + lff in one pass over the concat buffer
                 and os.environ.get("NVQ_FUSE_TAIL", "1") != "0")
    for k in range(nblocks):
        cat = cats[k]
        for i in range(LAYERS - 1 if fuse_tail else LAYERS):
            cin = F + GROWTH * i
            wp = packs.get(f"residual_blocks.{k}.layers.{i}.0.weight", False, cin)
            K.conv_forward(cat.inp(cin), wp, P[f"residual_blocks.{k}.layers.{i}.0.bias"],
                           cat.y(i), 3, relu=True, math=math,
                           bits=sv.bits[k][i] if use_bits else None, bits_mode=1 if use_bits else 0)
        wl = packs.get(f"residual_blocks.{k}.lff.weight", False, g.CAT)
        if fuse_tail:
            i = LAYERS - 1
            cin = F + GROWTH * i
            w3 = packs.get(f"residual_blocks.{k}.layers.{i}.0.weight", False, cin)
            K.rdb_tail_forward(cat.inp(cin), w3, P[f"residual_blocks.{k}.layers.{i}.0.bias"], cat.y(i), wl,
                               P[f"residual_blocks.{k}.lff.bias"], xloc(k + 1), alpha=0.2, res=cat.x(),
                               bits=sv.bits[k][i] if use_bits else None, tile_rows=_TAIL_ROWS)
        else:
            K.conv_forward(cat.inp(g.CAT), wl, P[f"residual_blocks.{k}.lff.bias"], xloc(k + 1), 1, alpha=0.2,
                           res=cat.x(), math=math)

    K.TIMER_TAG = ""
    # ---- global fusion + upsampler tail
    fused, gr = _new(dev, B, H, W, F, dtype=act_dtype), _new(dev, B, H, W, F, dtype=act_dtype)   # conv-to-conv tensors
    K.conv_forward(xloc(nblocks), packs.get("gff.0.weight", False, F), P["gff.0.bias"], Sl(fused), 3,
                   relu=True, out2=Sl(gr), res=center, math=math)
    out = _new(dev, B, g.Cimg, H * scale, W * scale)
    passmask = _new(dev, B, g.Cimg, H * scale, W * scale, dtype=torch.uint8)
    wup = packs.get("upsampler.conv.weight", False, F)
    if _fused_tail_ok(math, fused, g.Cimg, scale):
        # conv + pixel-shuffle (an LDS transpose in the conv's epilogue) + bicubic skip + clamp: one launch, no `u` tensor
        K.upsampler_tail_forward(Sl(fused), wup, P["upsampler.conv.bias"], frames, c, scale, out, passmask)
    else:
        u = _new(dev, B, H, W, g.Up)
        K.conv_forward(Sl(fused), wup, P["upsampler.conv.bias"], Sl(u, g.U), 3, cout_store=g.Up, math=math)
        K.shuffle_bicubic_clamp(u, frames, c, scale, out, passmask)
    sv.fused, sv.gr, sv.passmask = fused, gr, passmask
    sv.xloc = xloc
    return out, sv


def _wgrad(x: Sl, cin_w: int, dy: Sl, G: Dict[str, torch.Tensor], wname: str, bname: Optional[str], ws, ksize,
           alpha=1.0, math=K.MATH_F32):
    """ws: the shared workspace (reduce behind the kernel) or a _ReduceQueue (reduce deferred into its next batch)"""
    if isinstance(ws, _ReduceQueue):
        K.conv_wgrad(x, cin_w, dy, G[wname], G[bname] if bname else None, ws.ws(), ksize, alpha=alpha, math=math, defer=ws.jobs)
    else:
        K.conv_wgrad(x, cin_w, dy, G[wname], G[bname] if bname else None, ws, ksize, alpha=alpha, math=math)


def extract_features(P: Dict[str, torch.Tensor], frames: torch.Tensor, F: int, math: int = K.MATH_F32,
                     act_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Per-frame features of a whole video in eval mode (running BatchNorm statistics): frames (B,Tv,C,H,W) ->
    [Tv,B,H,W,F] fp32, to be handed to forward(features=...) window by window (EnhancementEngine.enhance_video)."""
    B, Tv, _, H, W = frames.shape
    out = _new(frames.device, Tv, B, H, W, F)
    for t0 in range(0, Tv, K.MAX_T):                         # the kernels take at most MAX_T frame groups per launch
        n = min(K.MAX_T, Tv - t0)
        part = frames[:, t0:t0 + n].contiguous()
        _extract(P, part, list(range(n)), F, False, math, act_dtype, Sl(out[t0:t0 + n].view(n * B, H, W, F)), n * B, None,
                 Saved())
    return out


def backward(P: Dict[str, torch.Tensor], sv: Saved, dout: torch.Tensor, G: Dict[str, torch.Tensor]) -> None:
    """Write the gradient of every parameter into G[name] (each exactly once, overwrite)."""
    if sv.feat0 is None:
        raise RuntimeError("forward(features=...) is an inference path: its result cannot be differentiated")
    g = sv.g
    dev = dout.device
    B, T, H, W, NI, NO, c, F = g.B, g.T, g.H, g.W, g.NI, g.NO, g.c, g.F
    math, act_dtype = sv.math, sv.act_dtype
    ws = workspace(dev)
    # Launch-bound steps (small frames: the 64x64 continual-learning step is ~330 dependent launches of a few microseconds): the
    # conv weight gradients leave their partial sums in workspaces of their own and one launch per eight of them does the reduces
    # (nvq_wgrad_reduce_batch; same sums, same order) - ~50 launches less per step.  Large frames keep the reduce behind each
    # kernel, where its partial slabs (up to 56 MB) are still in the last-level cache.
    small = g.B * g.H * g.W <= SMALL_STEP_PIXELS and K.TIMER is None
    wq = _ReduceQueue(dev) if small else ws
    nb = g.NB
    # the transposed packs of the input-gradient convs outside the dense blocks in one launch (the blocks' mirror-form packs
    # have a batched launch of their own below)
    packs = _Packs(P, math)
    pre_a = "temporal_aggregator.attention."
    reqs = [("upsampler.conv.weight", True, g.Up, F), ("gff.0.weight", True, F, F), (pre_a + "4.weight", True, g.Tp, F),
            (pre_a + "2.weight", True, F, F), (pre_a + "0.weight", True, F, T * F)]
    if NO:
        reqs += [(f"motion_estimator.flow_net.{idx}.weight", True, cs, keep)
                 for idx, cs, keep in zip((6, 4, 2, 0), (4, K.pad4(32), K.pad4(64), K.pad4(128)), (32, 64, 128, 81))]
    packs.prefetch(reqs)

    # ---- upsampler tail
    du = _new(dev, B, H, W, g.Up)
    K.shuffle_clamp_backward(dout, sv.passmask, g.s, du)
    _wgrad(Sl(sv.fused), F, Sl(du, g.U), G, "upsampler.conv.weight", "upsampler.conv.bias", wq, 3, math=math)
    # bf16 activation mode with reference frames: the feature gradient is FINISHED by the two correlation gradients, which write it
    # as bf16 (dfeat16, below).  Its terms then never meet in an fp32 accumulator: the upsampler input-gradient conv's second
    # output, the attention path's slice of daligned and the warp gradient are bf16 addends of those last passes (no fp32
    # gradient tensor, no axpy kernel, no read-modify-write).
    feat16 = (math == K.MATH_BF16 and act_dtype == torch.bfloat16 and F in (32, 64) and sv.img8 is not None and NO > 0
              and sv.aligned.dtype == torch.bfloat16 and os.environ.get("NVQ_BF16_FEATURE_GRAD", "1") != "0")
    # fp32 form: gradient w.r.t. the features of every frame (slot order); the centre frames' part is first written by the
    # upsampler's input-gradient conv (out2 below), the other frames' by the warp backward (overwrite mode), the rest added
    dfeat_all = _new(dev, B if feat16 else NI, 1 if feat16 else H, 1 if feat16 else W, F)   # (feat16: only a pointer for unused arguments)
    dfeat_c = dfeat_all[:B]
    dfeat_c16 = _new(dev, B, H, W, F, dtype=torch.bfloat16) if feat16 else None
    dg = _new(dev, B, H, W, F, dtype=act_dtype)
    K.conv_forward(Sl(du), packs.get("upsampler.conv.weight", True, g.Up, F), None, Sl(dg), 3,
                   out2=Sl(dfeat_c16) if feat16 else Sl(dfeat_c), mask=Sl(sv.gr), mask_c0=0, mask_c1=F, math=math)
    # ---- gff
    xN = sv.xloc(nb)
    _wgrad(xN, F, Sl(dg), G, "gff.0.weight", "gff.0.bias", wq, 3, math=math)
    # gradient buffers of the dense blocks, layout [gout(F) | dy_4 | dy_3 | dy_2 | dy_1 | dy_0] (ping-pong)
    dcats = [K.CatBuf(dev, B, H, W, F, LAYERS, g.CATLD, act_dtype, sv.planar) for _ in range(2)] if nb else []
    dagg = _new(dev, B, H, W, F, dtype=sv.cbam_dtype)
    gout = dcats[(nb - 1) & 1].x() if nb else Sl(dagg)
    K.conv_forward(Sl(dg), packs.get("gff.0.weight", True, F, F), None, gout, 3, math=math)
    _capture("dfused", (lambda: dfeat_c16.float()) if feat16 else dfeat_c)
    _capture("dres", lambda: gout.t[..., :F].float())

    # ---- residual dense blocks, last to first, in mirror form (see nvq_rdb_backward_weights)
    K.TIMER_TAG = "rdb"
    # the gout channels of the mirror-form convs only carry the 1x1 lff^T term: the bf16 kernels skip their other taps
    ctr = F if (math == K.MATH_BF16 and F % 32 == 0) else 0
    # the combined weights of all blocks, then their 6 * nb packs in one launch
    reqs = []
    for k in range(nb):
        pre = f"residual_blocks.{k}."
        wb, wbx = K.rdb_backward_weights(P[pre + "lff.weight"], [P[pre + f"layers.{i}.0.weight"] for i in range(LAYERS)], F)
        reqs += [(wb[j], False, F + GROWTH * j, None) for j in range(LAYERS)] + [(wbx, False, g.CAT, None)]
    mirror = K.conv_pack_many(reqs, math)
    for k in range(nb - 1, -1, -1):
        cat = sv.cats[k]
        dcat = dcats[k & 1]
        pre = f"residual_blocks.{k}."
        gout = dcat.x()
        _wgrad(cat.inp(g.CAT), g.CAT, gout, G, pre + "lff.weight", pre + "lff.bias", wq, 1, alpha=0.2, math=math)
        wpb = mirror[k * (LAYERS + 1):(k + 1) * (LAYERS + 1)]     # packs of Wb_4 .. Wb_0, Wb_x
        for i in range(LAYERS - 1, -1, -1):
            cinb = F + GROWTH * (LAYERS - 1 - i)            # channels [0, cinb) = gout, dy_4 .. dy_{i+1}
            dy = dcat.y(LAYERS - 1 - i)                      # slot of dy_i (channels [cinb, cinb + 32) of the buffer)
            if sv.bits is not None:
                K.conv_forward(dcat.inp(cinb), wpb[LAYERS - 1 - i], None, dy, 3,
                               math=math, bits=sv.bits[k][i], bits_mode=2, center_cin=ctr)
            else:
                K.conv_forward(dcat.inp(cinb), wpb[LAYERS - 1 - i], None, dy, 3,
                               mask=cat.y(i), mask_c0=0, mask_c1=GROWTH, math=math, center_cin=ctr)
            cin = F + GROWTH * i
            _wgrad(cat.inp(cin), cin, dy, G, pre + f"layers.{i}.0.weight", pre + f"layers.{i}.0.bias", wq, 3,
                   math=math)
        nxt = dcats[(k - 1) & 1].x() if k > 0 else Sl(dagg)
        K.conv_forward(dcat.inp(g.CAT), wpb[LAYERS], None, nxt, 3, res=gout, math=math, center_cin=ctr)
    K.TIMER_TAG = ""
    dprev = Sl(dagg)

    _capture("dagg", lambda: dprev.t[..., dprev.coff:dprev.coff + F].float())
    # ---- CBAM
    w1 = P["temporal_aggregator.refine.channel_attention.fc.0.weight"]
    w2 = P["temporal_aggregator.refine.channel_attention.fc.2.weight"]
    w7 = P["temporal_aggregator.refine.spatial_attention.conv.weight"]
    dpre = _new(dev, B, H, W)
    K.cbam_bwd_spatial_pre(dprev, sv.weighted, sv.ca, sv.sa, dpre)
    dsm = _new(dev, B, H, W, 2)
    K.cbam_bwd_spatial_conv(dpre, sv.sm, w7, dsm, G["temporal_aggregator.refine.spatial_attention.conv.weight"], ws)
    dweighted = _new(dev, B, H, W, F, dtype=sv.cbam_dtype)
    dca_partial = _new(dev, B, sv.nblk, F)
    K.cbam_bwd_scale(dprev, sv.weighted, sv.ca, sv.sa, dsm, sv.amax, dweighted, dca_partial)
    dgap_pix = _new(dev, B, F)
    K.cbam_bwd_channel(dca_partial, sv.nblk, F, g.R, B, H * W, w1, w2, sv.gap, sv.hid, sv.ca,
                       G["temporal_aggregator.refine.channel_attention.fc.0.weight"],
                       G["temporal_aggregator.refine.channel_attention.fc.2.weight"], dgap_pix)

    # ---- softmax-weighted sum and the attention convs
    daligned = _new(dev, B, H, W, T * F, dtype=sv.aligned.dtype)   # stored like `aligned` (bf16 in the bf16 activation mode)
    dlogits = _new(dev, B, H, W, g.Tp)
    K.tsum_backward(dweighted, dgap_pix, sv.aligned, sv.attn, T, F, daligned, dlogits)
    pre = "temporal_aggregator.attention."
    _wgrad(Sl(sv.a2), F, Sl(dlogits, T), G, pre + "4.weight", pre + "4.bias", wq, 3, math=math)
    da2 = _new(dev, B, H, W, F, dtype=act_dtype)
    K.conv_forward(Sl(dlogits), packs.get(pre + "4.weight", True, g.Tp, F), None, Sl(da2), 3,
                   mask=Sl(sv.a2), mask_c0=0, mask_c1=F, math=math)
    _wgrad(Sl(sv.a1), F, Sl(da2), G, pre + "2.weight", pre + "2.bias", wq, 3, math=math)
    da1 = _new(dev, B, H, W, F, dtype=act_dtype)
    if sv.a1_bits is not None:
        K.conv_forward(Sl(da2), packs.get(pre + "2.weight", True, F, F), None, Sl(da1), 3, math=math, bits=sv.a1_bits, bits_mode=2)
    else:
        K.conv_forward(Sl(da2), packs.get(pre + "2.weight", True, F, F), None, Sl(da1), 3,
                       mask=Sl(sv.a1), mask_c0=0, mask_c1=F, math=math)
    _wgrad(Sl(sv.aligned), T * F, Sl(da1), G, pre + "0.weight", pre + "0.bias", wq, 3, math=math)
    K.conv_forward(Sl(da1), packs.get(pre + "0.weight", True, F, T * F), None, Sl(daligned), 3,
                   accumulate=True, math=math)
    if not feat16:
        K.axpy_slice(Sl(dfeat_c), Sl(daligned, F, c * F))
    _capture("dweighted", lambda: dweighted.float())
    _capture("dlogits", dlogits)
    _capture("daligned", daligned)

    # ---- motion: warp, flow net, correlation
    if NO:
        # (feat16: the reference frames' term from the warp leaves as bf16 too and is an addend of the correlation gradient's pass)
        dfeat_oth = _new(dev, NO, H, W, F, dtype=torch.bfloat16) if feat16 else dfeat_all[B:]
        dflow = _new(dev, NO, H, W, 4)
        for j in range(1, T):
            t = g.slots[j]
            lo, hi = (j - 1) * B, j * B
            K.warp_backward(Sl(daligned, F, t * F), Sl(sv.feat_oth).images(lo, hi), sv.flow[lo:hi],
                            Sl(dfeat_oth).images(lo, hi), dflow[lo:hi], overwrite=True)
        acts = sv.flow_acts          # [corr(96), f1(128), f2(64), f3(32), flow(4)]
        chans = [81, 128, 64, 32, 2]
        dy_t, dy_c = dflow, 2
        for li, idx in reversed(list(enumerate((0, 2, 4, 6)))):
            x_t = acts[li]
            name = f"motion_estimator.flow_net.{idx}."
            x_sl = Sl(x_t) if li > 0 else Sl(x_t, CORR_LD, 0)
            _wgrad(x_sl, chans[li], Sl(dy_t, dy_c), G, name + "weight", name + "bias", wq, 3, math=math)
            cin_store = dy_t.shape[-1]
            wp = packs.get(name + "weight", True, cin_store, chans[li])
            dx_t = _new(dev, NO, H, W, x_t.shape[-1], dtype=act_dtype if li > 0 else x_t.dtype)
            if li > 0 and sv.flow_bits[li] is not None:
                K.conv_forward(Sl(dy_t), wp, None, Sl(dx_t, chans[li]), 3, math=math, bits=sv.flow_bits[li], bits_mode=2)
            elif li > 0:
                K.conv_forward(Sl(dy_t), wp, None, Sl(dx_t, chans[li]), 3, mask=Sl(x_t), mask_c0=0,
                               mask_c1=chans[li], math=math)
            else:
                # gradient of the correlation volume: the bf16 volume's 96-channel rows are written whole (zeros behind the
                # 81 real channels) - a row of 84 channels would leave lines half written (fill reads)
                full = dx_t.shape[-1] if dx_t.dtype == torch.bfloat16 else K.pad4(chans[li])
                K.conv_forward(Sl(dy_t), wp, None, Sl(dx_t, chans[li]), 3, cout_store=full, math=math)
            dy_t, dy_c = dx_t, chans[li]
        dcorr = dy_t
        _capture("f1", acts[1])
        _capture("dflow", dflow)
        _capture("dcorr", dcorr)
        center = Sl(sv.aligned, F, c * F)
        # The two correlation gradients are the LAST terms of the feature gradient.  bf16 activation mode: they write the finished
        # sum as bf16 (dfeat16) instead of back into the fp32 accumulator - the extractor's backward reads it three times
        # (BatchNorm sums, pointwise backward, the skip add of the first depthwise layer) at half the bytes, like every other
        # gradient it consumes.
        dfeat16 = _new(dev, NI, H, W, F, dtype=torch.bfloat16) if feat16 else None
        if feat16:
            K.correlation_backward(1, dcorr, center, Sl(dfeat_c), False, math=math, out16=dfeat16[B:], addends=(Sl(dfeat_oth),))
        else:
            K.correlation_backward(1, dcorr, center, Sl(dfeat_oth), True, math=math)
        # gradient w.r.t. the centre frame's features: the T - 1 reference frames in one pass
        if feat16:
            K.correlation_backward(2, dcorr, Sl(sv.feat_oth), Sl(dfeat_c), False, math=math, groups=T - 1, out16=dfeat16[:B],
                                   addends=(Sl(dfeat_c16), Sl(daligned, F, c * F)))
            dfeat_all = dfeat16
        else:
            K.correlation_backward(2, dcorr, Sl(sv.feat_oth), Sl(dfeat_c), True, math=math, groups=T - 1)

    # ---- feature extractor (all frames batched)
    _capture("dfeat_all", lambda: dfeat_all.float())
    dcur = dfeat_all
    bn_sums = None                    # BatchNorm-backward sums of layer k, when the depthwise backward of layer k + 1 left them
    # (built and measured, off by default: with the sums in it the depthwise backward only fits its registers with two instead
    # of five unrolled rows, and the step is 0.2 ms SLOWER than with the separate 0.54 ms reduce passes it replaces)
    fuse_sums = os.environ.get("NVQ_FUSED_BN_SUMS", "0") != "0"
    for k in (2, 1, 0):
        pre = f"feature_extractor.body.{k}."
        dd = _new(dev, NI, H, W, F, dtype=act_dtype)
        fused_bwd = (math == K.MATH_BF16 and act_dtype == torch.bfloat16 and F == 64 and sv.pws[k].dtype == torch.bfloat16
                     and sv.dws[k].dtype == torch.bfloat16)
        if fused_bwd and os.environ.get("NVQ_FUSED_PW_BWD", "1") != "0":
            # BatchNorm backward + the pointwise conv's input and weight gradients in one pass behind the BatchNorm sums: dp is
            # formed in LDS and never stored (nvq_pw_bn_backward)
            K.pw_bn_backward(dcur, sv.pws[k], sv.dws[k], B, sv.bn_mean[k], sv.bn_invstd[k], P[pre + "bn.weight"],
                             P[pre + "bn.bias"], sv.training, P[pre + "pointwise.weight"], dd, G[pre + "bn.weight"],
                             G[pre + "bn.bias"], G[pre + "pointwise.weight"], ws, sums_in=bn_sums)
            bn_sums = None
        else:
            dp = _new(dev, NI, H, W, F, dtype=act_dtype)
            K.bn_relu_backward(dcur, sv.pws[k], B, sv.bn_mean[k], sv.bn_invstd[k], P[pre + "bn.weight"],
                               P[pre + "bn.bias"], sv.training, dp, G[pre + "bn.weight"], G[pre + "bn.bias"], ws)
            _wgrad(Sl(sv.dws[k]), F, Sl(dp), G, pre + "pointwise.weight", None, wq, 1, math=math)
            K.conv_forward(Sl(dp), K.conv_pack(P[pre + "pointwise.weight"], True, F, F, math=math), None, Sl(dd), 1, math=math)
        xin, xin_bn = sv.dw_in[k]
        if fused_bwd and k > 0 and xin.dtype == torch.bfloat16 and os.environ.get("NVQ_FUSED_DW_BWD", "1") != "0":
            # the depthwise conv's input gradient and weight gradient from one staged tile (nvq_dwconv_backward).  Not for the
            # first layer: with the skip-path add and the head's ReLU mask in its epilogue the one-tile kernel takes as long as
            # the two launches below (2.53 vs 2.54 ms: the epilogue operands are loaded where they are used, by 8 waves per CU)
            dx = _new(dev, NI, H, W, F, dtype=act_dtype)
            prev = f"feature_extractor.body.{k - 1}."
            if fuse_sums and sv.training and os.environ.get("NVQ_FUSED_PW_BWD", "1") != "0":
                # ... and the backward sums of layer k - 1's BatchNorm: this kernel holds its input (xin) and the gradient of
                # its activation (dx), so the next pw_bn_backward needs no reduce pass of its own
                bn_sums = _new(dev, T, 2, F)
                K.dwconv_backward(xin, xin_bn, dd, P[pre + "depthwise.weight"], dx, G[pre + "depthwise.weight"], ws,
                                  bn_sums=bn_sums, bn_dgamma=G[prev + "bn.weight"], bn_dbeta=G[prev + "bn.bias"])
            else:
                K.dwconv_backward(xin, xin_bn, dd, P[pre + "depthwise.weight"], dx, G[pre + "depthwise.weight"], ws)
            dcur = dx
            continue
        K.dwconv_wgrad(xin, dd, G[pre + "depthwise.weight"], ws, bn=xin_bn)
        if k == 0 and sv.img8 is not None:
            # dx + skip path, ReLU-masked by the head features: the gradient of the head conv, as bf16
            dx = _new(dev, NI, H, W, F, dtype=act_dtype)
            K.dwconv_forward(dd, P[pre + "depthwise.weight"], dx, flip=True, add=dfeat_all, mask=sv.feat0)
        else:
            dx = _new(dev, NI, H, W, F, dtype=act_dtype if k > 0 else torch.float32)
            K.dwconv_forward(dd, P[pre + "depthwise.weight"], dx, flip=True)
        dcur = dx
    if sv.img8 is not None:
        K.conv_wgrad(Sl(sv.img8), g.Cimg, Sl(dcur), G["feature_extractor.head.0.weight"],
                     G["feature_extractor.head.0.bias"], ws, 3, math=math)
    else:
        # the skip path of  feat = body(h) + h  is summed inside the head kernel (dout2)
        K.head_wgrad(sv.frames, g.slots, dcur, sv.feat0, G["feature_extractor.head.0.weight"],
                     G["feature_extractor.head.0.bias"], ws, dout2=dfeat_all)
    if small:
        wq.flush()


# ----------------------------------------------------------------------------- LightweightSuperResolution
LIGHT_F = 32
LIGHT_BLOCKS = (2, 3, 4, 5)       # indices of the DepthwiseSeparableConv modules inside `net`

def light_forward(P: Dict[str, torch.Tensor], x: torch.Tensor, scale: int, training: bool, math: int = K.MATH_F32,
                  act_dtype: torch.dtype = torch.float32) -> "tuple[torch.Tensor, Saved]":
    """Single-frame net (reference super_resolution.py:434-470): conv3x3+ReLU, four depthwise-separable blocks,
    conv3x3 -> PixelShuffle, + bicubic(x), clamp.  Same kernels as the SR feature extractor and tail."""
    assert act_dtype == torch.float32 or math == K.MATH_BF16
    dev = x.device
    B, Cimg, H, W = x.shape
    F = LIGHT_F
    frames = x.view(B, 1, Cimg, H, W)
    ws = workspace(dev)
    sv = Saved()
    sv.frames, sv.training, sv.math, sv.act_dtype, sv.scale = frames, training, math, act_dtype, scale
    feat0 = _new(dev, B, H, W, F, dtype=act_dtype)
    K.head_forward(frames, [0], P["net.0.weight"], P["net.0.bias"], feat0, math=math)
    sv.feat0, sv.dws, sv.pws, sv.acts, sv.bn_mean, sv.bn_invstd = feat0, [], [], [], [], []
    cur = feat0
    for k in LIGHT_BLOCKS:
        pre = f"net.{k}."
        d = _new(dev, B, H, W, F, dtype=act_dtype)
        K.dwconv_forward(cur, P[pre + "depthwise.weight"], d)
        p = _new(dev, B, H, W, F, dtype=act_dtype)
        K.conv_forward(Sl(d), K.conv_pack(P[pre + "pointwise.weight"], False, F, math=math), None, Sl(p), 1, math=math)
        mean, invstd = _new(dev, 1, F), _new(dev, 1, F)
        if training:
            K.bn_stats(p, B, [0], mean, invstd, P[pre + "bn.running_mean"], P[pre + "bn.running_var"], ws, BN_EPS, BN_MOM)
            P[pre + "bn.num_batches_tracked"].add_(1)
        else:
            K.bn_eval_stats(P[pre + "bn.running_mean"], P[pre + "bn.running_var"], 1, mean, invstd, BN_EPS)
        r = _new(dev, B, H, W, F, dtype=act_dtype)
        K.bn_apply_relu(p, B, mean, invstd, P[pre + "bn.weight"], P[pre + "bn.bias"], None, Sl(r), B)
        sv.dws.append(d); sv.pws.append(p); sv.acts.append(r); sv.bn_mean.append(mean); sv.bn_invstd.append(invstd)
        cur = r
    U = Cimg * scale * scale
    Up = K.pad4(U)
    out = _new(dev, B, Cimg, H * scale, W * scale)
    passmask = _new(dev, B, Cimg, H * scale, W * scale, dtype=torch.uint8)
    wup = K.conv_pack(P["net.6.weight"], False, F, math=math)
    if _fused_tail_ok(math, cur, Cimg, scale):
        K.upsampler_tail_forward(Sl(cur), wup, P["net.6.bias"], frames, 0, scale, out, passmask)
    else:
        u = _new(dev, B, H, W, Up)
        K.conv_forward(Sl(cur), wup, P["net.6.bias"], Sl(u, U), 3, cout_store=Up, math=math)
        K.shuffle_bicubic_clamp(u, frames, 0, scale, out, passmask)
    sv.passmask, sv.U, sv.Up = passmask, U, Up
    return out, sv


def light_backward(P: Dict[str, torch.Tensor], sv: Saved, dout: torch.Tensor, G: Dict[str, torch.Tensor]) -> None:
    dev = dout.device
    B, _, Cimg, H, W = sv.frames.shape
    F, math, act_dtype = LIGHT_F, sv.math, sv.act_dtype
    ws = workspace(dev)
    du = _new(dev, B, H, W, sv.Up)
    K.shuffle_clamp_backward(dout, sv.passmask, sv.scale, du)
    last = sv.acts[-1]
    _wgrad(Sl(last), F, Sl(du, sv.U), G, "net.6.weight", "net.6.bias", ws, 3, math=math)
    dcur = _new(dev, B, H, W, F)
    K.conv_forward(Sl(du), K.conv_pack(P["net.6.weight"], True, sv.Up, F, math=math), None, Sl(dcur), 3, math=math)
    for j in range(len(LIGHT_BLOCKS) - 1, -1, -1):
        pre = f"net.{LIGHT_BLOCKS[j]}."
        dp = _new(dev, B, H, W, F, dtype=act_dtype)
        K.bn_relu_backward(dcur, sv.pws[j], B, sv.bn_mean[j], sv.bn_invstd[j], P[pre + "bn.weight"], P[pre + "bn.bias"],
                           sv.training, dp, G[pre + "bn.weight"], G[pre + "bn.bias"], ws)
        _wgrad(Sl(sv.dws[j]), F, Sl(dp), G, pre + "pointwise.weight", None, ws, 1, math=math)
        dd = _new(dev, B, H, W, F, dtype=act_dtype)
        K.conv_forward(Sl(dp), K.conv_pack(P[pre + "pointwise.weight"], True, F, F, math=math), None, Sl(dd), 1, math=math)
        xin = sv.feat0 if j == 0 else sv.acts[j - 1]
        K.dwconv_wgrad(xin, dd, G[pre + "depthwise.weight"], ws)
        dx = _new(dev, B, H, W, F, dtype=act_dtype if j > 0 else torch.float32)
        K.dwconv_forward(dd, P[pre + "depthwise.weight"], dx, flip=True)
        dcur = dx
    K.head_wgrad(sv.frames, [0], dcur, sv.feat0, G["net.0.weight"], G["net.0.bias"], ws)


def nhwc_to_nchw(t: torch.Tensor, c: Optional[int] = None, coff: int = 0) -> torch.Tensor:
    c = t.shape[-1] - coff if c is None else c
    return t[..., coff:coff + c].permute(0, 3, 1, 2).contiguous().float()


def intermediates(sv: Saved) -> dict:
    """return_intermediate payload in the reference's NCHW layout (super_resolution.py:384-389)."""
    g = sv.g
    feats: List[Optional[torch.Tensor]] = [None] * g.T
    aligned: List[Optional[torch.Tensor]] = [None] * g.T
    for j, t in enumerate(g.slots):
        if j == 0:
            feats[t] = nhwc_to_nchw(sv.aligned, g.F, g.c * g.F)
        else:
            feats[t] = nhwc_to_nchw(sv.feat_oth[(j - 1) * g.B:j * g.B])
        aligned[t] = nhwc_to_nchw(sv.aligned, g.F, t * g.F)
    x0 = sv.xloc(0)
    return {"features": feats, "aligned": aligned, "aggregated": nhwc_to_nchw(x0.t, g.F, 0)}
