"""ctypes binding of libnvq.so (include/nvq.h).

This is the only place that touches the C ABI.  There is no CPU or PyTorch fallback:
if the shared library is missing, or a tensor is not a contiguous fp32 HIP tensor, the
call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVQ_LIB", os.path.join(os.path.dirname(_HERE), "libnvq.so"))

MATH_F32 = 0
MATH_BF16 = 1
MAX_T = 8

_lib = None

vp, ci, cl, cf, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [
        ("inp", vp), ("in_ld", ci), ("in_coff", ci), ("cin", ci),
        ("wpack", vp),
        ("bias", vp),
        ("out", vp), ("out_ld", ci), ("out_coff", ci), ("cout", ci),
        ("cout_store", ci),
        ("out2", vp), ("out2_ld", ci), ("out2_coff", ci),
        ("res", vp), ("res_ld", ci), ("res_coff", ci), ("res_cmax", ci),
        ("mask", vp), ("mask_ld", ci), ("mask_coff", ci), ("mask_c0", ci), ("mask_c1", ci),
        ("n", ci), ("h", ci), ("w", ci),
        ("ksize", ci),
        ("relu", ci),
        ("alpha", cf),
        ("accumulate", ci),
        ("math", ci),
        ("in_bf16", ci), ("out_bf16", ci), ("out2_bf16", ci), ("res_bf16", ci), ("mask_bf16", ci),
        ("bits", vp), ("bits_mode", ci), ("center_cin", ci), ("in_plane", C.c_uint),
        ("tile_rows", ci), ("bits_words", ci),
    ]


class DwEpilogue(C.Structure):
    _fields_ = [("add", vp), ("add_ld", ci), ("mask", vp), ("mask_ld", ci), ("mask_bf16", ci), ("add_bf16", ci)]


class CorrAddends(C.Structure):
    _fields_ = [("a", vp), ("a_ld", ci), ("a_coff", ci), ("b", vp), ("b_ld", ci), ("b_coff", ci)]


class BnInput(C.Structure):
    _fields_ = [("mean", vp), ("invstd", vp), ("gamma", vp), ("beta", vp), ("group_images", ci)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", vp), ("x_ld", ci), ("x_coff", ci), ("cin", ci), ("cin_w", ci),
        ("dy", vp), ("dy_ld", ci), ("dy_coff", ci), ("cout", ci),
        ("dw", vp), ("dbias", vp),
        ("workspace", vp), ("workspace_bytes", sz),
        ("n", ci), ("h", ci), ("w", ci), ("ksize", ci),
        ("alpha", cf), ("accumulate", ci), ("math", ci),
        ("x_bf16", ci), ("dy_bf16", ci), ("x_plane", C.c_uint), ("variant", ci),
    ]


class WgradReduceJob(C.Structure):
    _fields_ = [
        ("part", vp), ("bias_part", vp), ("dw", vp), ("dbias", vp),
        ("nsplit", ci), ("nci", ci), ("nco", ci), ("taps", ci), ("cout", ci), ("cin_w", ci),
        ("alpha", cf), ("accumulate", ci),
    ]


_IP = C.POINTER(ci)

# name -> (restype, argtypes); must list every symbol declared in include/nvq.h
SIGNATURES = {
    "nvq_version": (ci, []),
    "nvq_last_error": (C.c_char_p, []),
    "nvq_conv_pack_floats": (sz, [ci, ci, ci, ci]),
    "nvq_conv_pack": (ci, [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "nvq_conv_forward": (ci, [C.POINTER(ConvDesc), vp]),
    "nvq_rdb_tail_forward": (ci, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), vp]),
    "nvq_rdb_backward_weights_floats": (sz, [ci]),
    "nvq_rdb_backward_weights": (ci, [vp, vp, vp, vp, vp, vp, ci, vp, vp]),
    "nvq_sizeof_conv_desc": (sz, []),
    "nvq_wgrad_workspace_bytes": (sz, []),
    "nvq_conv_wgrad": (ci, [C.POINTER(WgradDesc), vp]),
    "nvq_conv_wgrad_partial": (ci, [C.POINTER(WgradDesc), C.POINTER(WgradReduceJob), vp]),
    "nvq_wgrad_reduce_batch": (ci, [C.POINTER(WgradReduceJob), ci, vp]),
    "nvq_sizeof_wgrad_desc": (sz, []),
    "nvq_sizeof_wgrad_reduce_job": (sz, []),
    "nvq_conv_pack_batch": (ci, [vp, ci, ci, vp]),
    "nvq_head_forward": (ci, [vp, ci, ci, ci, ci, ci, _IP, ci, vp, vp, ci, vp, ci, ci, vp, ci, vp]),
    "nvq_head_wgrad": (ci, [vp, ci, ci, ci, ci, ci, _IP, ci, vp, ci, vp, ci, vp, ci, ci, vp, vp, vp, sz, ci, ci, vp]),
    "nvq_dwconv_forward": (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]),
    "nvq_dwconv_wgrad": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, sz, ci, ci, ci, vp, vp]),
    "nvq_bn_stats": (ci, [vp, ci, ci, ci, ci, ci, ci, cf, cf, _IP, vp, vp, vp, vp, vp, sz, ci, vp]),
    "nvq_bn_eval_stats": (ci, [vp, vp, ci, ci, cf, vp, vp, vp]),
    "nvq_bn_apply_relu": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, ci, vp, ci, ci, ci, ci, ci, vp]),
    "nvq_bn_relu_backward": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, sz, ci, ci, ci, ci, vp]),
    "nvq_dwconv_backward": (ci, [vp, ci, vp, vp, ci, vp, vp, ci, vp, ci, ci, ci, vp, vp, vp, vp, vp, sz, vp]),
    "nvq_dwpw_forward": (ci, [vp, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, cf, cf, _IP, vp, vp, vp, vp, vp, sz, vp]),
    "nvq_pw_bn_backward": (ci, [vp, ci, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, sz, vp]),
    "nvq_correlation_forward": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, vp]),
    "nvq_correlation_backward": (ci, [ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp, vp]),
    "nvq_warp_forward": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, vp]),
    "nvq_warp_backward": (ci, [vp, ci, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, vp, sz, ci, ci, ci, ci, vp]),
    "nvq_tsum_blocks": (ci, [ci, ci]),
    "nvq_tsum_forward": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, ci, vp, ci, ci, vp]),
    "nvq_tsum_backward": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, ci, ci, vp]),
    "nvq_cbam_channel": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    "nvq_cbam_pool": (ci, [vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, vp]),
    "nvq_cbam_spatial_apply": (ci, [vp, ci, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, vp]),
    "nvq_cbam_bwd_spatial_pre": (ci, [vp, ci, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_cbam_bwd_spatial_conv": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp, sz, ci, vp]),
    "nvq_cbam_bwd_scale": (ci, [vp, ci, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp, ci, vp, ci, vp]),
    "nvq_cbam_bwd_channel": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    "nvq_upsampler_tail_forward": (ci, [C.POINTER(ConvDesc), vp, ci, ci, ci, ci, vp, vp, vp]),
    "nvq_shuffle_bicubic_clamp": (ci, [vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]),
    "nvq_shuffle_clamp_backward": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_bicubic_blend": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, vp, vp]),
    "nvq_nchw_to_nhwc": (ci, [vp, cl, ci, ci, ci, ci, vp, ci, ci, ci, vp]),
    "nvq_nhwc_to_nchw": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, cl, vp]),
    "nvq_bn2_workspace_bytes": (sz, [ci]),
    "nvq_bn2_stats": (ci, [vp, ci, ci, cl, cf, cf, vp, vp, vp, vp, vp, sz, ci, vp]),
    "nvq_bn2_eval_stats": (ci, [vp, vp, ci, cf, vp, vp, vp]),
    "nvq_bn2_apply": (ci, [vp, ci, ci, cl, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, vp]),
    "nvq_bn2_backward": (ci, [vp, ci, vp, ci, ci, cl, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, vp, ci, vp, vp, vp, sz, ci, vp]),
    "nvq_maxpool_forward": (ci, [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp]),
    "nvq_maxpool_backward": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_subsample2": (ci, [vp, ci, ci, ci, ci, vp, ci, ci, vp]),
    "nvq_bilinear_resize": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_convt_pack": (ci, [vp, ci, ci, vp, vp]),
    "nvq_convt_unpack_grad": (ci, [vp, ci, ci, vp, vp]),
    "nvq_depth_space2": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "nvq_cast_slice": (ci, [vp, ci, ci, ci, vp, ci, ci, ci, ci, cl, cf, ci, vp]),
    "nvq_tconv_relayout": (ci, [vp, vp, ci, ci, ci, vp]),
    "nvq_tconv_cat": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "nvq_tconv_grad_combine": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
    "nvq_gap_blocks": (ci, [ci, ci]),
    "nvq_gap_partial": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "nvq_add_image_channel": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "nvq_tanh": (ci, [vp, vp, cl, vp, ci, vp]),
    "nvq_fusion_mix_forward": (ci, [vp, vp, ci, vp, ci, vp, ci, ci, cl, vp, vp, vp, vp]),
    "nvq_fusion_mix_backward": (ci, [vp, vp, vp, ci, cl, vp, ci, vp, ci, vp, ci, vp]),
    "nvq_mask_blend": (ci, [vp, vp, ci, vp, ci, ci, ci, ci, vp, vp]),
    "nvq_mask_blend_backward": (ci, [vp, vp, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_stem7_forward": (ci, [vp, vp, ci, ci, ci, ci, vp, ci, ci, vp]),
    "nvq_stem7_wgrad": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, sz, ci, vp]),
    "nvq_pixel_shuffle": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp]),
    "nvq_mse_forward": (ci, [vp, vp, cl, vp, vp, sz, vp]),
    "nvq_mse_backward": (ci, [vp, vp, cl, vp, vp, vp]),
    "nvq_axpy_slice": (ci, [vp, ci, ci, vp, ci, ci, vp, ci, ci, ci, cl, cf, ci, ci, vp]),
    "nvq_colsum": (ci, [vp, ci, ci, ci, cl, cf, vp, vp, sz, ci, vp]),
    "nvq_ewc_penalty": (ci, [vp, vp, vp, cl, cf, vp, vp, sz, vp]),
    "nvq_ewc_penalty_grad": (ci, [vp, vp, vp, cl, cf, vp, vp, ci, vp]),
    "nvq_fisher_accumulate": (ci, [vp, cl, vp, vp]),
    "nvq_si_update": (ci, [vp, vp, cl, vp, vp, vp]),
    "nvq_si_consolidate": (ci, [vp, cl, cf, vp, vp, vp, vp]),
}


def lib():
    """Load libnvq.so once; raise loudly when it is absent (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libnvq.so not found at {LIB_PATH}: build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "nerve_cl has no CPU / PyTorch fallback for the super-resolution hot path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        assert l.nvq_sizeof_conv_desc() == C.sizeof(ConvDesc), "nvq_conv_desc layout mismatch"
        assert l.nvq_sizeof_wgrad_desc() == C.sizeof(WgradDesc), "nvq_wgrad_desc layout mismatch"
        assert l.nvq_sizeof_wgrad_reduce_job() == C.sizeof(WgradReduceJob), "nvq_wgrad_reduce_job layout mismatch"
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {lib().nvq_last_error().decode()}")


def stream() -> vp:
    return vp(torch.cuda.current_stream().cuda_stream)


def require_device(t: torch.Tensor, what: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} lives on {t.device}: the nerve_cl super-resolution path runs only as HIP kernels "
            "on an AMD GPU (move the model and its inputs to 'cuda'); there is no CPU fallback.")


def device_guard(dev: torch.device):
    """Context that makes `dev` the current HIP device: libnvq launches on the CURRENT device's stream, so a tensor on
    cuda:1 must not be handed to a launch issued while cuda:0 is current."""
    import contextlib
    return torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()


def is_bf16(t: Optional[torch.Tensor]) -> int:
    return int(t is not None and t.dtype == torch.bfloat16)


def ptr(t: Optional[torch.Tensor], off: int = 0) -> Optional[int]:
    """Device address of element `off` of a contiguous fp32/int32/uint8 tensor (None passes NULL)."""
    if t is None:
        return None
    require_device(t)
    assert t.is_contiguous(), "libnvq needs contiguous tensors"
    return t.data_ptr() + off * t.element_size()


def int_array(vals: Sequence[int]):
    return (ci * len(vals))(*vals)


class Sl:
    """A channel slice of an NHWC activation buffer (fp32, or bf16 for conv-internal tensors):
    tensor [N,H,W,ld], channels [coff, coff+c); ld / coff count elements of the tensor's dtype."""

    __slots__ = ("t", "ld", "coff", "c", "plane")

    def __init__(self, t: torch.Tensor, c: Optional[int] = None, coff: int = 0, plane: int = 0):
        """plane != 0 (conv / weight-gradient INPUTS only): `t` is the leading tensor of a slice-planar buffer (CatBuf) and
        the c channels continue in the compact 32-channel planes behind it (nvq_conv_desc::in_plane)."""
        assert t.dtype in (torch.float32, torch.bfloat16) and t.dim() == 4 and t.is_contiguous()
        self.t, self.ld, self.coff, self.plane = t, t.shape[-1], coff, plane
        self.c = t.shape[-1] - coff if c is None else c
        assert self.coff + self.c <= self.ld or (plane and coff == 0)

    @property
    def n(self):
        return self.t.shape[0]

    @property
    def bf16(self) -> int:
        return int(self.t.dtype == torch.bfloat16)

    def images(self, lo: int, hi: int) -> "Sl":
        assert not self.plane
        return Sl(self.t[lo:hi], self.c, self.coff)

    def base(self) -> int:
        """address of channel `coff` of pixel 0"""
        return ptr(self.t, self.coff)


class CatBuf:
    """The buffer of one residual dense block: [x (F channels) | 32-channel slices ...] (forward: x, y_0 .. y_4; backward:
    gout, dy_4 .. dy_0).  Interleaved: one [N,H,W,ld] tensor, a slice = a channel range (torch.cat for free).  Slice-planar
    (planar=True, bf16): x and every slice are compact tensors of their own in one allocation, so a layer's 64-byte pixel
    rows fill whole 128-byte lines; the convs / weight gradients that read a channel prefix take it through `inp()`."""

    def __init__(self, dev, N: int, H: int, W: int, F: int, nslices: int, ld: int, dtype, planar: bool):
        self.F, self.planar = F, planar
        if planar:
            assert dtype == torch.bfloat16 and F in (32, 64, 128)
            npx = N * H * W
            self.plane = npx * 32
            self.flat = torch.empty((F // 32 + nslices) * self.plane, dtype=dtype, device=dev)
            self.lead = self.flat[:npx * F].view(N, H, W, F)
            self.slices = [self.flat[(F // 32 + j) * self.plane:(F // 32 + j + 1) * self.plane].view(N, H, W, 32)
                           for j in range(nslices)]
        else:
            self.t = torch.empty((N, H, W, ld), dtype=dtype, device=dev)

    def x(self) -> Sl:
        return Sl(self.lead) if self.planar else Sl(self.t, self.F, 0)

    def y(self, j: int) -> Sl:
        return Sl(self.slices[j]) if self.planar else Sl(self.t, 32, self.F + 32 * j)

    def inp(self, cin: int) -> Sl:
        return Sl(self.lead, cin, 0, plane=self.plane) if self.planar else Sl(self.t, cin, 0)


class KernelTimer:
    """HIP-event timing of individual launches on the current stream (bench.py only)."""

    def __init__(self):
        self.records = []          # (label, flops, bytes, ev0, ev1, shape)
        self.tags = []             # section tag of each record (TIMER_TAG at launch time)
        self.enabled = True

    def start(self):
        if not self.enabled:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, ev0, label, flops, nbytes, shape=""):
        if ev0 is None:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        self.records.append((label, flops, nbytes, ev0, ev1, shape))
        self.tags.append(TIMER_TAG)

    def _durations(self):
        """ms per record.  An event pair brackets the launch on the stream, so when the GPU has caught up with the host the
        pair also contains the host's launch gap; such records (more than twice the median of their (label, shape) group)
        are counted at the group median.  Returns (durations, number of records replaced)."""
        raw = [e0.elapsed_time(e1) for _, _, _, e0, e1, _ in self.records]
        groups = {}
        for i, (label, _, _, _, _, shape) in enumerate(self.records):
            groups.setdefault((label, shape), []).append(i)
        fixed = 0
        for idx in groups.values():
            vals = sorted(raw[i] for i in idx)
            med = vals[len(vals) // 2]
            for i in idx:
                if raw[i] > 2.0 * med:
                    raw[i] = med
                    fixed += 1
        self.outliers = fixed
        return raw

    def summary(self):
        """{label: dict(launches, ms_total, flops, bytes)} (call after a device synchronize)."""
        out = {}
        dur = self._durations()
        for (label, fl, nb, _, _, _), ms in zip(self.records, dur):
            d = out.setdefault(label, dict(launches=0, ms_total=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms_total"] += ms
            d["flops"] += fl
            d["bytes"] += nb
        return out

    def by_tag(self):
        """{tag: dict(launches, ms_total, flops, bytes)} over the engine's section tags (e.g. "rdb")."""
        out = {}
        dur = self._durations()
        for (label, fl, nb, _, _, _), ms, tag in zip(self.records, dur, self.tags):
            d = out.setdefault(tag, dict(launches=0, ms_total=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms_total"] += ms
            d["flops"] += fl
            d["bytes"] += nb
        return out

    def by_shape(self):
        """{(label, shape): dict(...)} for the per-shape table of bench.py --detail."""
        out = {}
        dur = self._durations()
        for (label, fl, nb, _, _, shape), ms in zip(self.records, dur):
            d = out.setdefault((label, shape), dict(launches=0, ms_total=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms_total"] += ms
            d["flops"] += fl
            d["bytes"] += nb
        return out


TIMER: Optional[KernelTimer] = None
TIMER_TAG = ""     # set by the engine around a section (bench.py's per-section roofline)


def _nt(cout: int) -> int:
    return 16 if cout <= 16 else (32 if cout <= 32 else 64)


def wgrad_workspace_bytes() -> int:
    return int(lib().nvq_wgrad_workspace_bytes())


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


# ----------------------------------------------------------------------------- convolution
class PackJob(C.Structure):
    """nvq_pack_job"""
    _fields_ = [("w", vp), ("cout_w", ci), ("cin_w", ci), ("ksize", ci), ("transpose", ci), ("cin_store", ci),
                ("cout_keep", ci), ("wpack", vp)]


def conv_pack_many(reqs, math: int = MATH_F32) -> "list[torch.Tensor]":
    """conv_pack of every (w, transpose, cin_store, cout_keep) of `reqs` in one launch per 48 jobs (nvq_conv_pack_batch);
    the results are slices of one buffer."""
    if not reqs:
        return []
    sizes, ws = [], []
    for w, transpose, cin_store, cout_keep in reqs:
        cout_w, cin_w, k, _ = w.shape
        keep = (cin_w if cout_keep is None else cout_keep) if transpose else cout_w
        sizes.append((int(lib().nvq_conv_pack_floats(keep, cin_store, k, math)), keep))
        ws.append(w.contiguous())
    flat = torch.empty(sum((n + 3) // 4 * 4 for n, _ in sizes), dtype=torch.float32, device=ws[0].device)
    jobs = (PackJob * len(reqs))()
    outs, off = [], 0
    for i, ((w, transpose, cin_store, _), (n, keep)) in enumerate(zip(reqs, sizes)):
        out = flat[off:off + n]
        off += (n + 3) // 4 * 4
        cout_w, cin_w, k, _ = w.shape
        jobs[i] = PackJob(ptr(ws[i]), cout_w, cin_w, k, int(transpose), cin_store, keep if transpose else 0, ptr(out))
        outs.append(out)
    check(lib().nvq_conv_pack_batch(C.cast(jobs, vp), len(reqs), math, stream()), "nvq_conv_pack_batch")
    return outs


def conv_pack(w: torch.Tensor, transpose: bool, cin_store: int, cout_keep: Optional[int] = None,
              math: int = MATH_F32) -> torch.Tensor:
    """Pack a PyTorch conv weight [Cout, Cin, k, k] for nvq_conv_forward (mode-specific layout)."""
    cout_w, cin_w, k, _ = w.shape
    keep = (cin_w if cout_keep is None else cout_keep) if transpose else cout_w
    n = lib().nvq_conv_pack_floats(keep, cin_store, k, math)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib().nvq_conv_pack(ptr(w.contiguous()), cout_w, cin_w, k, int(transpose), cin_store,
                              keep if transpose else 0, math, ptr(wp), stream()), "nvq_conv_pack")
    return wp


def _conv_desc(x: Sl, wpack: torch.Tensor, bias: Optional[torch.Tensor], out: Sl, ksize: int, *,
                 relu: bool = False, alpha: float = 1.0, accumulate: bool = False,
                 cout_store: Optional[int] = None, out2: Optional[Sl] = None,
                 res: Optional[Sl] = None, mask: Optional[Sl] = None, mask_c0: int = 0,
                 mask_c1: int = 0, math: int = MATH_F32, alg_cin: Optional[int] = None,
                 bits: Optional[torch.Tensor] = None, bits_mode: int = 0, center_cin: int = 0,
                 tile_rows: int = 0) -> ConvDesc:
    """bits: int32 [N,H,W] or [N,H,W,words] one-bit ReLU masks (see nvq_conv_desc): bits_mode 1 = write, 2 = read as the mask.
    center_cin: leading input channels whose weights are zero outside the centre tap (a hint, see nvq_conv_desc)."""
    n, h, w, _ = x.t.shape
    assert out.t.shape[:3] == x.t.shape[:3]
    d = ConvDesc()
    d.inp, d.in_ld, d.in_coff, d.cin = ptr(x.t), x.ld, x.coff, x.c
    d.wpack = ptr(wpack)
    d.bias = ptr(bias)
    d.out, d.out_ld, d.out_coff, d.cout = ptr(out.t), out.ld, out.coff, out.c
    d.cout_store = out.c if cout_store is None else cout_store
    if out2 is not None:
        d.out2, d.out2_ld, d.out2_coff = ptr(out2.t), out2.ld, out2.coff
    if res is not None:
        d.res, d.res_ld, d.res_coff, d.res_cmax = ptr(res.t), res.ld, res.coff, res.c
    if mask is not None:
        d.mask, d.mask_ld, d.mask_coff, d.mask_c0, d.mask_c1 = ptr(mask.t), mask.ld, mask.coff, mask_c0, mask_c1
    d.n, d.h, d.w, d.ksize = n, h, w, ksize
    d.relu, d.alpha, d.accumulate, d.math = int(relu), alpha, int(accumulate), math
    d.in_bf16, d.out_bf16 = x.bf16, out.bf16
    d.out2_bf16 = out2.bf16 if out2 is not None else 0
    d.res_bf16 = res.bf16 if res is not None else 0
    d.mask_bf16 = mask.bf16 if mask is not None else 0
    if bits_mode:
        assert bits is not None and bits.dtype == torch.int32 and tuple(bits.shape[:3]) == (n, h, w) and bits.is_contiguous()
        d.bits, d.bits_mode = ptr(bits), bits_mode
        d.bits_words = bits.shape[3] if bits.dim() == 4 else 1        # [N,H,W] (cout <= 32) or [N,H,W,ceil(cout / 32)]
    d.center_cin = center_cin
    d.in_plane = x.plane
    d.tile_rows = tile_rows
    return d


def conv_forward(x: Sl, wpack: torch.Tensor, bias: Optional[torch.Tensor], out: Sl, ksize: int, *,
                 relu: bool = False, alpha: float = 1.0, accumulate: bool = False,
                 cout_store: Optional[int] = None, out2: Optional[Sl] = None,
                 res: Optional[Sl] = None, mask: Optional[Sl] = None, mask_c0: int = 0,
                 mask_c1: int = 0, math: int = MATH_F32, alg_cin: Optional[int] = None,
                 bits: Optional[torch.Tensor] = None, bits_mode: int = 0, center_cin: int = 0, tile_rows: int = 0) -> None:
    n, h, w, _ = x.t.shape
    ev0 = TIMER.start() if TIMER is not None else None
    d = _conv_desc(x, wpack, bias, out, ksize, relu=relu, alpha=alpha, accumulate=accumulate, cout_store=cout_store,
                   out2=out2, res=res, mask=mask, mask_c0=mask_c0, mask_c1=mask_c1, math=math, bits=bits,
                   bits_mode=bits_mode, center_cin=center_cin, tile_rows=tile_rows)
    check(lib().nvq_conv_forward(C.byref(d), stream()), "nvq_conv_forward")
    if ev0 is not None:
        cin = x.c if alg_cin is None else alg_cin
        esz = lambda sl: 2.0 if sl.bf16 else 4.0   # noqa: E731  bytes per element
        nbytes = cin * esz(x) + out.c * esz(out) * (2 if accumulate else 1) \
            + (res.c * esz(res) if res is not None else 0) + (out2.c * esz(out2) if out2 is not None else 0) \
            + ((mask_c1 - mask_c0) * esz(mask) if mask is not None else 0) + (4 * (bits.shape[3] if bits is not None and bits.dim() == 4 else 1) if bits_mode else 0)
        TIMER.stop(ev0, f"conv_{'bf16' if math == MATH_BF16 else 'f32'}_kernel<{_nt(out.c) // 16},{ksize}>", 2.0 * n * h * w * (cin * ksize * ksize - center_cin * (ksize * ksize - 1)) * out.c,
                   n * h * w * nbytes, f"n{n} cin{x.c}{'h' if x.bf16 else ''} cout{out.c}{'h' if out.bf16 else ''}"
                   + (" acc" if accumulate else "")
                   + (" res" if res is not None else "") + (" mask" if mask is not None else "")
                   + (" bitsW" if bits_mode == 1 else " bitsR" if bits_mode == 2 else "")
                   + (f" ctr{center_cin}" if center_cin else ""))


def rdb_tail_forward(x: Sl, w3: torch.Tensor, b3, y4: Sl, wl: torch.Tensor, bl, out: Sl, *, alpha: float, res: Sl,
                     bits: Optional[torch.Tensor] = None, tile_rows: int = 0) -> None:
    """Last dense layer (3x3, x -> y4 = the next 32 channels of the same buffer, bias + ReLU) and the block's 1x1 fusion
    over [x | y4] (-> out = alpha * (lff + bias) + res) in one launch (nvq_rdb_tail_forward).  tile_rows = 4: the four-wave
    kernel instead of the eight-wave, two-role one (same results)."""
    n, h, w, _ = x.t.shape
    ev0 = TIMER.start() if TIMER is not None else None
    d3 = _conv_desc(x, w3, b3, y4, 3, relu=True, math=MATH_BF16, bits=bits, bits_mode=1 if bits is not None else 0,
                    tile_rows=tile_rows)
    xl = Sl(x.t, x.c + y4.c, x.coff, plane=x.plane)
    dl = _conv_desc(xl, wl, bl, out, 1, alpha=alpha, res=res, math=MATH_BF16)
    check(lib().nvq_rdb_tail_forward(C.byref(d3), C.byref(dl), stream()), "nvq_rdb_tail_forward")
    if ev0 is not None:
        esz = lambda sl: 2.0 if sl.bf16 else 4.0   # noqa: E731
        TIMER.stop(ev0, "rdb_tail_kernel", 2.0 * n * h * w * (x.c * y4.c * 9 + xl.c * out.c),
                   n * h * w * (x.c * 2.0 + y4.c * 2.0 + res.c * esz(res) + out.c * esz(out)),
                   f"n{n} cin{x.c}h -> 32 + lff {xl.c} -> {out.c}{'h' if out.bf16 else ''}")


def rdb_backward_weights(lff: torch.Tensor, ws: Sequence[torch.Tensor], F: int):
    """Combined 'mirror' weights of one dense block: returns ([Wb_4, Wb_3, Wb_2, Wb_1, Wb_0], Wb_x) as views
    of one buffer, PyTorch layout (see nvq_rdb_backward_weights)."""
    n = lib().nvq_rdb_backward_weights_floats(F)
    out = torch.empty(n, dtype=torch.float32, device=lff.device)
    check(lib().nvq_rdb_backward_weights(ptr(lff), *[ptr(w) for w in ws], F, ptr(out), stream()),
          "nvq_rdb_backward_weights")
    views, off = [], 0
    for t in range(5):
        k = 32 * (F + 32 * t) * 9
        views.append(out[off:off + k].view(32, F + 32 * t, 3, 3))
        off += k
    return views, out[off:].view(F, F + 160, 3, 3)


def conv_wgrad(x: Sl, cin_w: int, dy: Sl, dw: torch.Tensor, dbias: Optional[torch.Tensor],
               ws: torch.Tensor, ksize: int, *, alpha: float = 1.0, accumulate: bool = False,
               math: int = MATH_F32, variant: int = 0, defer: Optional[list] = None) -> None:
    """defer: a list -> only the weight-gradient kernel runs (nvq_conv_wgrad_partial); its reduce job is appended to the list
    and finished by wgrad_reduce_batch(list).  `ws` must then be this call's own until that batch has run."""
    n, h, w, _ = x.t.shape
    ev0 = TIMER.start() if TIMER is not None and defer is None else None
    d = WgradDesc()
    d.x, d.x_ld, d.x_coff, d.cin, d.cin_w = ptr(x.t), x.ld, x.coff, x.c, cin_w
    d.dy, d.dy_ld, d.dy_coff, d.cout = ptr(dy.t), dy.ld, dy.coff, dy.c
    d.dw, d.dbias = ptr(dw), ptr(dbias)
    d.workspace, d.workspace_bytes = ptr(ws), ws.numel() * ws.element_size()
    d.n, d.h, d.w, d.ksize = n, h, w, ksize
    d.alpha, d.accumulate, d.math = alpha, int(accumulate), math
    d.x_bf16, d.dy_bf16, d.x_plane = x.bf16, dy.bf16, x.plane
    d.variant = variant
    if defer is not None:
        job = WgradReduceJob()
        check(lib().nvq_conv_wgrad_partial(C.byref(d), C.byref(job), stream()), "nvq_conv_wgrad_partial")
        defer.append((job, (ws, dw, dbias)))               # (the tensors stay alive with the job)
        return
    check(lib().nvq_conv_wgrad(C.byref(d), stream()), "nvq_conv_wgrad")
    if ev0 is not None:
        TIMER.stop(ev0, f"wgrad_{'bf16' if math == MATH_BF16 else 'f32'}_kernel<{ksize}>", 2.0 * n * h * w * cin_w * dy.c * ksize * ksize,
                   n * h * w * (cin_w * (2.0 if x.bf16 else 4.0) + dy.c * (2.0 if dy.bf16 else 4.0)),
                   f"n{n} cin{cin_w}{'h' if x.bf16 else ''} cout{dy.c}{'h' if dy.bf16 else ''}")


def wgrad_reduce_batch(jobs: list) -> None:
    """Finish the weight gradients deferred into `jobs` (conv_wgrad(..., defer=jobs)): up to 16 reduces per launch."""
    if not jobs:
        return
    arr = (WgradReduceJob * len(jobs))(*[j for j, _ in jobs])
    check(lib().nvq_wgrad_reduce_batch(arr, len(jobs), stream()), "nvq_wgrad_reduce_batch")
    jobs.clear()


# ----------------------------------------------------------------------------- feature extractor
def head_forward(frames: torch.Tensor, slots: Sequence[int], weight, bias, out: torch.Tensor,
                 img8: Optional[torch.Tensor] = None, math: int = MATH_F32) -> None:
    B, T, Cin, H, W = frames.shape
    F = weight.shape[0]
    assert img8 is None or (img8.dtype == torch.bfloat16 and tuple(img8.shape) == (len(slots) * B, H, W, 8))
    check(lib().nvq_head_forward(ptr(frames), B, T, Cin, H, W, int_array(slots), len(slots), ptr(weight),
                                 ptr(bias), F, ptr(out), out.shape[-1], is_bf16(out), ptr(img8), math, stream()),
          "nvq_head_forward")


def head_wgrad(frames, slots, dout: torch.Tensor, act: torch.Tensor, dweight, dbias, ws, accumulate=False,
               dout2: Optional[torch.Tensor] = None):
    B, T, Cin, H, W = frames.shape
    F = dweight.shape[0]
    check(lib().nvq_head_wgrad(ptr(frames), B, T, Cin, H, W, int_array(slots), len(slots), ptr(dout),
                               dout.shape[-1], ptr(dout2), dout2.shape[-1] if dout2 is not None else 0, ptr(act),
                               act.shape[-1], F, ptr(dweight), ptr(dbias), ptr(ws),
                               ws.numel() * 4, int(accumulate), is_bf16(act), stream()), "nvq_head_wgrad")


def _bn_input(bn) -> "Optional[BnInput]":
    """bn = (mean, invstd, gamma, beta, group_images) or None."""
    if bn is None:
        return None
    b = BnInput()
    b.mean, b.invstd, b.gamma, b.beta, b.group_images = ptr(bn[0]), ptr(bn[1]), ptr(bn[2]), ptr(bn[3]), int(bn[4])
    return b


def dwconv_bn_fusable(x: torch.Tensor, C: int) -> bool:
    """True when the depthwise kernels can evaluate relu(bn(x)) while staging x (bf16 tensor, C % 64 == 0)."""
    return x.dtype == torch.bfloat16 and C % 64 == 0


def dwconv_forward(x: torch.Tensor, weight, out: torch.Tensor, flip=False, bn=None,
                   add: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None):
    """add (fp32) / mask (fp32 or bf16), both [N,H,W,>=C]: out = (conv + add) where mask > 0 (64-channel bf16 kernels)."""
    N, H, W, ld = x.shape
    Cc = weight.shape[0]
    b = _bn_input(bn)
    e = None
    if add is not None or mask is not None:
        e = DwEpilogue()
        e.add, e.add_ld, e.add_bf16 = ptr(add), add.shape[-1] if add is not None else 0, is_bf16(add)
        e.mask, e.mask_ld, e.mask_bf16 = ptr(mask), mask.shape[-1] if mask is not None else 0, is_bf16(mask)
    check(lib().nvq_dwconv_forward(ptr(x), ld, ptr(weight), Cc, ptr(out), out.shape[-1], N, H, W, int(flip),
                                   is_bf16(x), is_bf16(out), C.byref(b) if b is not None else None,
                                   C.byref(e) if e is not None else None, stream()),
          "nvq_dwconv_forward")


def dwconv_wgrad(x: torch.Tensor, dy: torch.Tensor, dweight, ws, accumulate=False, bn=None):
    N, H, W, ld = x.shape
    Cc = dweight.shape[0]
    b = _bn_input(bn)
    check(lib().nvq_dwconv_wgrad(ptr(x), ld, ptr(dy), dy.shape[-1], Cc, N, H, W, ptr(dweight), ptr(ws),
                                 ws.numel() * 4, int(accumulate), is_bf16(x), is_bf16(dy),
                                 C.byref(b) if b is not None else None, stream()), "nvq_dwconv_wgrad")


def bn_stats(x: torch.Tensor, group_images: int, order: Sequence[int], mean, invstd, rmean, rvar, ws,
             eps=1e-5, momentum=0.1):
    N, H, W, ld = x.shape
    Cc = mean.shape[-1]
    check(lib().nvq_bn_stats(ptr(x), ld, Cc, N, group_images, H, W, eps, momentum, int_array(order), ptr(mean),
                             ptr(invstd), ptr(rmean), ptr(rvar), ptr(ws), ws.numel() * 4, is_bf16(x), stream()),
          "nvq_bn_stats")


def bn_eval_stats(rmean, rvar, G: int, mean, invstd, eps=1e-5):
    check(lib().nvq_bn_eval_stats(ptr(rmean), ptr(rvar), rmean.numel(), G, eps, ptr(mean), ptr(invstd), stream()),
          "nvq_bn_eval_stats")


def bn_apply_relu(x: torch.Tensor, group_images: int, mean, invstd, gamma, beta, res: Optional[torch.Tensor],
                  outA: Sl, split_images: int, outB: Optional[Sl] = None):
    N, H, W, ld = x.shape
    Cc = gamma.numel()
    check(lib().nvq_bn_apply_relu(ptr(x), ld, Cc, N, group_images, H, W, ptr(mean), ptr(invstd), ptr(gamma),
                                  ptr(beta), ptr(res), res.shape[-1] if res is not None else 0, ptr(outA.t),
                                  outA.ld, outA.coff, split_images, ptr(outB.t) if outB else None,
                                  outB.ld if outB else 0, outB.coff if outB else 0, is_bf16(x), outA.bf16, is_bf16(res),
                                  stream()),
          "nvq_bn_apply_relu")


def bn_relu_backward(dy: torch.Tensor, x: torch.Tensor, group_images: int, mean, invstd, gamma, beta,
                     training: bool, dx: torch.Tensor, dgamma, dbeta, ws, accumulate=False):
    N, H, W, ld = x.shape
    Cc = gamma.numel()
    check(lib().nvq_bn_relu_backward(ptr(dy), dy.shape[-1], ptr(x), ld, Cc, N, group_images, H, W, ptr(mean),
                                     ptr(invstd), ptr(gamma), ptr(beta), int(training), ptr(dx), dx.shape[-1],
                                     ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel() * 4, int(accumulate), is_bf16(dy),
                                     is_bf16(x), is_bf16(dx), stream()),
          "nvq_bn_relu_backward")


def dwconv_backward(x: torch.Tensor, bn, dy: torch.Tensor, weight: torch.Tensor, dx: torch.Tensor, dweight, ws,
                    add: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, bn_sums=None, bn_dgamma=None,
                    bn_dbeta=None) -> None:
    """Input and weight gradient of a 64-channel depthwise 3x3 conv from one staged tile (bf16 mode): see nvq_dwconv_backward.
    bn = (mean, invstd, gamma, beta, group_images): the conv was fed relu(bn(x)); add (fp32) / mask (bf16): dx = (dx + add)
    where mask > 0.  bn_sums [G, 2, 64] (+ bn_dgamma, bn_dbeta [64]): also return the backward sums of that BatchNorm (its input is
    x, the gradient of its activation is dx) - pw_bn_backward(..., sums_in=bn_sums) then skips its reduce pass."""
    N, H, W, ld = x.shape
    assert x.dtype == dy.dtype == dx.dtype == torch.bfloat16 and weight.shape[0] == 64
    b = _bn_input(bn)
    e = None
    if add is not None or mask is not None:
        e = DwEpilogue()
        e.add, e.add_ld, e.add_bf16 = ptr(add), add.shape[-1] if add is not None else 0, is_bf16(add)
        e.mask, e.mask_ld, e.mask_bf16 = ptr(mask), mask.shape[-1] if mask is not None else 0, is_bf16(mask)
    check(lib().nvq_dwconv_backward(ptr(x), ld, C.byref(b) if b is not None else None, ptr(dy), dy.shape[-1],
                                    ptr(weight.contiguous()), ptr(dx), dx.shape[-1], C.byref(e) if e is not None else None,
                                    N, H, W, ptr(dweight), ptr(bn_sums), ptr(bn_dgamma), ptr(bn_dbeta), ptr(ws), ws.numel() * 4,
                                    stream()), "nvq_dwconv_backward")


def dwpw_forward(x: torch.Tensor, bn, dw_weight: torch.Tensor, pw_weight: torch.Tensor, d: torch.Tensor, p: torch.Tensor,
                 group_images: int, order: Optional[Sequence[int]], mean, invstd, rmean, rvar, ws, eps=1e-5,
                 momentum=0.1) -> None:
    """depthwise 3x3 -> pointwise 1x1 -> BatchNorm statistics in one pass (bf16 mode, 64 channels): see nvq_dwpw_forward.
    bn = (mean, invstd, gamma, beta, group_images) of the previous layer or None; order None: no statistics (eval mode)."""
    N, H, W, ld = x.shape
    assert x.dtype == d.dtype == p.dtype == torch.bfloat16 and tuple(pw_weight.shape[:2]) == (64, 64)
    b = _bn_input(bn)
    stats = order is not None
    ev0 = TIMER.start() if TIMER is not None else None
    check(lib().nvq_dwpw_forward(ptr(x), ld, C.byref(b) if b is not None else None, ptr(dw_weight.contiguous()),
                                 ptr(pw_weight.contiguous()), ptr(d), d.shape[-1], ptr(p), p.shape[-1], N, group_images, H, W,
                                 int(stats), eps, momentum, int_array(order) if stats else None, ptr(mean) if stats else None,
                                 ptr(invstd) if stats else None, ptr(rmean) if stats else None, ptr(rvar) if stats else None,
                                 ptr(ws), ws.numel() * 4, stream()), "nvq_dwpw_forward")
    if ev0 is not None:
        npx = N * H * W
        TIMER.stop(ev0, "dwpw_fwd_kernel", 2.0 * npx * 64 * (64 + 9), npx * 64 * 2.0 * 3, f"n{N} 64->64")


def pw_bn_backward(dy: torch.Tensor, p: torch.Tensor, d: torch.Tensor, group_images: int, mean, invstd, gamma, beta,
                   training: bool, weight: torch.Tensor, dd: torch.Tensor, dgamma, dbeta, dweight, ws, sums_in=None) -> None:
    """Backward of pointwise conv -> BatchNorm -> ReLU in one pass (bf16 mode, 64 channels): see nvq_pw_bn_backward."""
    N, H, W, _ = p.shape
    assert p.dtype == d.dtype == dd.dtype == torch.bfloat16 and tuple(weight.shape[:2]) == (64, 64)
    ev0 = TIMER.start() if TIMER is not None else None
    check(lib().nvq_pw_bn_backward(ptr(dy), dy.shape[-1], is_bf16(dy), ptr(p), p.shape[-1], ptr(d), d.shape[-1], N,
                                   group_images, H, W, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), int(training),
                                   ptr(weight.contiguous()), ptr(dd), dd.shape[-1], ptr(dgamma), ptr(dbeta), ptr(dweight),
                                   ptr(sums_in), ptr(ws), ws.numel() * 4, stream()), "nvq_pw_bn_backward")
    if ev0 is not None:
        npx = N * H * W
        TIMER.stop(ev0, "pw_bn_bwd_kernel", 2.0 * npx * 64 * 64 * 2,
                   npx * 64 * (2.0 * 2 + (2 if is_bf16(dy) else 4) * 2 + 2), f"n{N} 64->64 {'h' if is_bf16(dy) else 'f'}")


# ----------------------------------------------------------------------------- motion
def correlation_forward(x1: Sl, x2: Sl, out: torch.Tensor, math: int = MATH_F32):
    """x1 / x2: fp32, or (MATH_BF16, C in {32, 64}) both bf16-stored feature tensors."""
    N, H, W, _ = x1.t.shape
    assert x1.bf16 == x2.bf16
    check(lib().nvq_correlation_forward(x1.base(), x1.ld, x2.base(), x2.ld, x2.n, x1.c, N, H, W, ptr(out),
                                        out.shape[-1], math, is_bf16(out), x1.bf16, stream()), "nvq_correlation_forward")


def correlation_backward(which: int, dcorr: torch.Tensor, other: Sl, dx: Sl, accumulate: bool, math: int = MATH_F32,
                         groups: int = 1, out16: Optional[torch.Tensor] = None, addends: Sequence[Sl] = ()):
    """groups > 1 (which == 2): dcorr / other hold groups * N images, frame-major; dx (N images) collects all of them in
    one pass.  out16 (bf16 [N/groups, H, W, >= C]): the finished gradient is written there as bf16 instead of back to dx;
    addends (with out16): up to two bf16 slices added in the same pass (accumulate False: dx is not read)."""
    N, H, W, ld = dcorr.shape
    assert N % groups == 0 and (out16 is None or out16.dtype == torch.bfloat16) and len(addends) <= 2
    ad = None
    if addends:
        assert out16 is not None and all(a.bf16 for a in addends)
        ad = CorrAddends()
        ad.a, ad.a_ld, ad.a_coff = ptr(addends[0].t), addends[0].ld, addends[0].coff
        if len(addends) > 1:
            ad.b, ad.b_ld, ad.b_coff = ptr(addends[1].t), addends[1].ld, addends[1].coff
    check(lib().nvq_correlation_backward(which, ptr(dcorr), ld, other.base(), other.ld, other.n, other.c, N // groups, H, W,
                                         ptr(dx.t), dx.ld, dx.coff, int(accumulate), math, is_bf16(dcorr), other.bf16,
                                         groups, ptr(out16), out16.shape[-1] if out16 is not None else 0,
                                         C.byref(ad) if ad is not None else None, stream()),
          "nvq_correlation_backward")


def warp_forward(feat: Sl, flow: torch.Tensor, out: Sl):
    N, H, W, _ = feat.t.shape
    check(lib().nvq_warp_forward(feat.base(), feat.ld, ptr(flow), flow.shape[-1], feat.c, N, H, W, ptr(out.t),
                                 out.ld, out.coff, feat.bf16, out.bf16, stream()), "nvq_warp_forward")


def warp_backward(dout: Sl, feat: Sl, flow: torch.Tensor, dfeat: Sl, dflow: torch.Tensor, gather: bool = True,
                  overwrite: bool = False):
    """gather=True: atomics-free two-pass form (needs 20 B of scratch per pixel, allocated here); False: scatter form.
    overwrite: dfeat is written instead of added to (gather form): no zero fill before, no read-modify-write here."""
    N, H, W, _ = feat.t.shape
    rec = torch.empty(N * H * W * 5 + 4, dtype=torch.float32, device=flow.device) if gather else None
    check(lib().nvq_warp_backward(ptr(dout.t), dout.ld, dout.coff, feat.base(), feat.ld, ptr(flow),
                                  flow.shape[-1], feat.c, N, H, W, dfeat.base(), dfeat.ld, ptr(dflow),
                                  dflow.shape[-1], ptr(rec), rec.numel() * 4 if rec is not None else 0, feat.bf16, dout.bf16,
                                  int(overwrite), dfeat.bf16, stream()),
          "nvq_warp_backward")


# ----------------------------------------------------------------------------- aggregation
def tsum_blocks(H: int, W: int) -> int:
    return int(lib().nvq_tsum_blocks(H, W))


def tsum_forward(aligned: torch.Tensor, logits: torch.Tensor, T: int, Cc: int, attn, weighted, gap_partial):
    N, H, W, ld = aligned.shape
    check(lib().nvq_tsum_forward(ptr(aligned), ld, ptr(logits), logits.shape[-1], T, Cc, N, H, W, ptr(attn),
                                 attn.shape[-1], ptr(weighted), weighted.shape[-1], ptr(gap_partial), is_bf16(aligned),
                                 is_bf16(weighted), stream()),
          "nvq_tsum_forward")


def tsum_backward(dweighted, dgap_pix, aligned, attn, T: int, Cc: int, daligned, dlogits):
    N, H, W, ld = aligned.shape
    check(lib().nvq_tsum_backward(ptr(dweighted), dweighted.shape[-1], ptr(dgap_pix), ptr(aligned), ld, ptr(attn),
                                  attn.shape[-1], T, Cc, N, H, W, ptr(daligned), daligned.shape[-1], ptr(dlogits),
                                  dlogits.shape[-1], is_bf16(aligned), is_bf16(daligned), is_bf16(dweighted), stream()),
          "nvq_tsum_backward")


def cbam_channel(gap_partial, nblk, Cc, R, N, HW, w1, w2, gap, hid, ca):
    check(lib().nvq_cbam_channel(ptr(gap_partial), nblk, Cc, R, N, HW, ptr(w1), ptr(w2), ptr(gap), ptr(hid),
                                 ptr(ca), stream()), "nvq_cbam_channel")


def cbam_pool(x: torch.Tensor, ca, sm, amax):
    N, H, W, ld = x.shape
    check(lib().nvq_cbam_pool(ptr(x), ld, ptr(ca), ca.shape[-1], N, H, W, ptr(sm), ptr(amax), is_bf16(x), stream()),
          "nvq_cbam_pool")


def cbam_spatial_apply(x: torch.Tensor, ca, sm, w7, sa, out: Sl):
    N, H, W, ld = x.shape
    check(lib().nvq_cbam_spatial_apply(ptr(x), ld, ptr(ca), ptr(sm), ptr(w7), ca.shape[-1], N, H, W, ptr(sa),
                                       ptr(out.t), out.ld, out.coff, out.bf16, is_bf16(x), stream()), "nvq_cbam_spatial_apply")


def cbam_bwd_spatial_pre(dout: Sl, x: torch.Tensor, ca, sa, dpre):
    N, H, W, ld = x.shape
    assert dout.bf16 == is_bf16(x), "dout and x are stored alike"
    check(lib().nvq_cbam_bwd_spatial_pre(ptr(dout.t), dout.ld, dout.coff, ptr(x), ld, ptr(ca), ptr(sa),
                                         ca.shape[-1], N, H, W, ptr(dpre), is_bf16(x), stream()), "nvq_cbam_bwd_spatial_pre")


def cbam_bwd_spatial_conv(dpre, sm, w7, dsm, dw7, ws, accumulate=False):
    N, H, W = dpre.shape[:3]
    check(lib().nvq_cbam_bwd_spatial_conv(ptr(dpre), ptr(sm), ptr(w7), N, H, W, ptr(dsm), ptr(dw7), ptr(ws),
                                          ws.numel() * 4, int(accumulate), stream()), "nvq_cbam_bwd_spatial_conv")


def cbam_bwd_scale(dout: Sl, x: torch.Tensor, ca, sa, dsm, amax, dx: torch.Tensor, dca_partial):
    N, H, W, ld = x.shape
    assert dout.bf16 == is_bf16(x) == is_bf16(dx), "dout, x and dx are stored alike"
    check(lib().nvq_cbam_bwd_scale(ptr(dout.t), dout.ld, dout.coff, ptr(x), ld, ptr(ca), ptr(sa), ptr(dsm),
                                   ptr(amax), ca.shape[-1], N, H, W, ptr(dx), dx.shape[-1], ptr(dca_partial),
                                   is_bf16(x), stream()), "nvq_cbam_bwd_scale")


def cbam_bwd_channel(dca_partial, nblk, Cc, R, N, HW, w1, w2, gap, hid, ca, dw1, dw2, dgap_pix, accumulate=False):
    check(lib().nvq_cbam_bwd_channel(ptr(dca_partial), nblk, Cc, R, N, HW, ptr(w1), ptr(w2), ptr(gap), ptr(hid),
                                     ptr(ca), ptr(dw1), ptr(dw2), ptr(dgap_pix), int(accumulate), stream()),
          "nvq_cbam_bwd_channel")


# ----------------------------------------------------------------------------- upsampler tail
def upsampler_tail_forward(x: Sl, wpack: torch.Tensor, bias: torch.Tensor, frames: torch.Tensor, t_center: int, s: int, out,
                           passmask) -> None:
    """conv3x3(x -> Cimg*s*s) + PixelShuffle(s) + bicubic skip + clamp in one launch (bf16 mode, bf16 x)."""
    B, T, Cimg, H, W = frames.shape
    n, h, w, _ = x.t.shape
    assert (n, h, w) == (B, H, W)
    ev0 = TIMER.start() if TIMER is not None else None
    d = ConvDesc()
    d.inp, d.in_ld, d.in_coff, d.cin = ptr(x.t), x.ld, x.coff, x.c
    d.wpack, d.bias = ptr(wpack), ptr(bias)
    d.cout = d.cout_store = Cimg * s * s
    d.n, d.h, d.w, d.ksize, d.alpha, d.math, d.in_bf16 = n, h, w, 3, 1.0, MATH_BF16, x.bf16
    check(lib().nvq_upsampler_tail_forward(C.byref(d), ptr(frames), T, t_center, Cimg, s, ptr(out), ptr(passmask), stream()),
          "nvq_upsampler_tail_forward")
    if ev0 is not None:
        U = Cimg * s * s
        TIMER.stop(ev0, "upsampler_tail_kernel", 2.0 * n * h * w * x.c * 9 * U,
                   n * h * w * (x.c * 2.0 + Cimg * 4.0 + Cimg * s * s * 5.0), f"n{n} cin{x.c}h s{s}")


def shuffle_bicubic_clamp(u: torch.Tensor, frames: torch.Tensor, t_center: int, s: int, out, passmask):
    B, T, Cimg, H, W = frames.shape
    check(lib().nvq_shuffle_bicubic_clamp(ptr(u), u.shape[-1], ptr(frames), B, T, t_center, Cimg, H, W, s,
                                          ptr(out), ptr(passmask), stream()), "nvq_shuffle_bicubic_clamp")


def shuffle_clamp_backward(dout: torch.Tensor, passmask, s: int, du: torch.Tensor):
    B, Cimg, OH, OW = dout.shape
    check(lib().nvq_shuffle_clamp_backward(ptr(dout), ptr(passmask), B, Cimg, OH // s, OW // s, s, ptr(du),
                                           du.shape[-1], stream()), "nvq_shuffle_clamp_backward")


def bicubic_blend(sr: torch.Tensor, frames: torch.Tensor, t_center: int, s: int, strength: float, out: torch.Tensor):
    B, T, Cimg, H, W = frames.shape
    check(lib().nvq_bicubic_blend(ptr(sr), ptr(frames), B, T, t_center, Cimg, H, W, s, float(strength), ptr(out),
                                  stream()), "nvq_bicubic_blend")


# ----------------------------------------------------------------------------- helpers
def axpy_slice(dst: Sl, src: Sl, alpha: float = 1.0, accumulate: bool = True, mask: Optional[Sl] = None):
    npix = dst.t.shape[0] * dst.t.shape[1] * dst.t.shape[2]
    assert dst.c == src.c
    check(lib().nvq_axpy_slice(ptr(dst.t), dst.ld, dst.coff, ptr(src.t), src.ld, src.coff,
                               ptr(mask.t) if mask else None, mask.ld if mask else 0, mask.coff if mask else 0,
                               dst.c, npix, alpha, int(accumulate), src.bf16, stream()), "nvq_axpy_slice")


def colsum(x: Sl, out: torch.Tensor, ws, alpha=1.0, accumulate=False):
    npix = x.t.shape[0] * x.t.shape[1] * x.t.shape[2]
    check(lib().nvq_colsum(ptr(x.t), x.ld, x.coff, x.c, npix, alpha, ptr(out), ptr(ws), ws.numel() * 4,
                           int(accumulate), stream()), "nvq_colsum")


# ----------------------------------------------------------------------------- loss
def mse_forward(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, ws: torch.Tensor):
    check(lib().nvq_mse_forward(ptr(a), ptr(b), a.numel(), ptr(out), ptr(ws), ws.numel() * 4, stream()), "nvq_mse_forward")


def mse_backward(a: torch.Tensor, b: torch.Tensor, go_dev: Optional[torch.Tensor], da: torch.Tensor):
    check(lib().nvq_mse_backward(ptr(a), ptr(b), a.numel(), ptr(go_dev), ptr(da), stream()), "nvq_mse_backward")


# ----------------------------------------------------------------------------- EWC
def ewc_penalty(theta, star, fisher, lam: float, out, ws):
    check(lib().nvq_ewc_penalty(ptr(theta), ptr(star), ptr(fisher), theta.numel(), lam, ptr(out), ptr(ws),
                                ws.numel() * 4, stream()), "nvq_ewc_penalty")


def ewc_penalty_grad(theta, star, fisher, lam: float, scale_dev, grad, accumulate: bool):
    check(lib().nvq_ewc_penalty_grad(ptr(theta), ptr(star), ptr(fisher), theta.numel(), lam, ptr(scale_dev),
                                     ptr(grad), int(accumulate), stream()), "nvq_ewc_penalty_grad")


def si_update(theta, grad, p_old, W):
    check(lib().nvq_si_update(ptr(theta), ptr(grad), theta.numel(), ptr(p_old), ptr(W), stream()), "nvq_si_update")


def si_consolidate(theta, damping: float, p_old, W, omega):
    check(lib().nvq_si_consolidate(ptr(theta), theta.numel(), damping, ptr(p_old), ptr(W), ptr(omega), stream()),
          "nvq_si_consolidate")


def fisher_accumulate(grad, fisher):
    check(lib().nvq_fisher_accumulate(ptr(grad), grad.numel(), ptr(fisher), stream()), "nvq_fisher_accumulate")
