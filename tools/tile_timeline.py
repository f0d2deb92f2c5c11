"""GPU box (diagnostic build): per-workgroup timeline of one dense-layer conv launch (conv_m32_kernel, cin <= 128): every
workgroup's wave 0 stamps s_memrealtime (100 MHz) at entry / first fetch issued / first chunk staged / K loop done / epilogue
arithmetic done / stores issued, plus its HW_ID and XCC_ID; this script groups the workgroups by CU and prints where a tile's
time goes and how long a CU's workgroup slot stays empty between two workgroups.  NVQ_DEBUG_TOOLS=1 bash build.sh first."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NVQ_LIB", os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd", "libnvq_debug.so"))
sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import numpy as np
import torch
from nerve_cl import _nvq as K

dev = torch.device("cuda", 0)
N, H, W, F = int(os.environ.get("TL_N", 8)), 540, 960, 64
cin = int(os.environ.get("TL_CIN", 128))
torch.manual_seed(0)
cat = K.CatBuf(dev, N, H, W, F, 5, F + 160, torch.bfloat16, True)
cat.flat.copy_(torch.randn(cat.flat.numel(), device=dev).clamp_(-3, 3).to(torch.bfloat16))
w = torch.randn(32, cin, 3, 3, device=dev) * 0.05
b = torch.randn(32, device=dev)
wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
yi = (cin - F) // 32
ntiles = ((W + 31) // 32) * ((H + 15) // 16) * N
stamps = torch.zeros(ntiles * 16, dtype=torch.int32, device=dev)
d = K._conv_desc(cat.inp(cin), wp, b, cat.y(yi), 3, relu=True, math=K.MATH_BF16)
for mode in (0, 8):
    K.lib().nvq_debug_set_conv_mode(mode)
    if mode == 8:
        d.bits = stamps.data_ptr()
    for _ in range(3):
        K.check(K.lib().nvq_conv_forward(C.byref(d), K.stream()), "nvq_conv_forward")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    K.check(K.lib().nvq_conv_forward(C.byref(d), K.stream()), "nvq_conv_forward")
    e1.record(); torch.cuda.synchronize()
    print(f"mode {mode}: launch {e0.elapsed_time(e1) * 1e3:.1f} us ({ntiles} tiles)")
K.lib().nvq_debug_set_conv_mode(0)
s = stamps.cpu().numpy().astype(np.uint32).reshape(ntiles, 16)
t = (s[:, 0:12:2].astype(np.uint64) | (s[:, 1:12:2].astype(np.uint64) << np.uint64(32))).astype(np.int64)
t = (t - t[:, 0].min()) / 100.0                           # us since the first workgroup's entry
hw, xcc = s[:, 12], s[:, 13] & 0xf
cu = (xcc.astype(np.int64) << 8) | ((hw >> 8) & 0xff)
names = ["entry -> first fetch issued", "-> first chunk in LDS", "-> K loop done", "-> epilogue arithmetic done", "-> stores issued"]
for i, nm in enumerate(names):
    dt = t[:, i + 1] - t[:, i]
    print(f"{nm:32s} median {np.median(dt):6.2f} us   p10 {np.percentile(dt, 10):6.2f}   p90 {np.percentile(dt, 90):6.2f}")
dur = t[:, 5] - t[:, 0]
print(f"tile (entry -> stores issued)    median {np.median(dur):6.2f} us; launch span {t[:, 5].max():.1f} us; CUs seen {len(set(cu.tolist()))}")
gaps, occ = [], []
for c in sorted(set(cu.tolist())):
    idx = np.where(cu == c)[0]
    st, en = np.sort(t[idx, 0]), np.sort(t[idx, 5])
    # a slot frees at an end time; the next start at or after it is its refill
    for e in en:
        later = st[st >= e]
        if later.size: gaps.append(later[0] - e)
    span = en.max() - st.min()
    occ.append((t[idx, 5] - t[idx, 0]).sum() / span)
gaps = np.array(gaps)
print(f"slot refill (a workgroup's last stamp -> the next entry on that CU): median {np.median(gaps):5.2f} us  p90 {np.percentile(gaps, 90):5.2f}")
print(f"workgroups resident per CU, time average: {np.mean(occ):.2f} (2 = both slots always taken);  workgroups per CU {ntiles / len(set(cu.tolist())):.1f}")
first = np.sort(t[:, 0])
print(f"entry times: first 512 workgroups within {first[min(511, ntiles - 1)]:.1f} us; last entry {first[-1]:.1f} us")
