"""GPU box, with a library built from an INSTRUMENTED copy of csrc/conv_bf16.hip (cycle stamps in the 16x32-tile 3x3 kernel,
written through a __device__ pointer set by nvq_debug_set_stamps; not shipped): where does a tile's time go?
Stamps: 0 kernel entry, 1 index arithmetic done, 2 first fetch issued (+ centre-tap chunks), 3 first full chunk committed and
the next fetch issued, 4 K loop done, 5 epilogue barrier passed, 6 epilogue arithmetic + LDS staging done, 7 stores issued,
8 stores retired.   NVQ_LIB=<instrumented .so> python tools/conv_stamps.py"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd"))
import numpy as np
import torch
from nerve_cl import _nvq as K

N, H, W = 8, 540, 960
stamps = torch.zeros(4 * 8 * 16, dtype=torch.int64, device="cuda")
L = ctypes.CDLL(K.LIB_PATH)
L.nvq_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert L.nvq_debug_set_stamps(stamps.data_ptr()) == 0
names = ["index math", "fetch(0) issue", "first chunk wait+commit", "rest of K loop", "epilogue barrier", "epilogue math+stage",
         "store issue", "store retire"]
for cin, bits_mode in [(64, 1), (192, 1), (192, 2)]:
    x = torch.randn(N, H, W, cin, device="cuda").bfloat16()              # compact (planar-like) input
    out = torch.empty(N, H, W, 32, device="cuda", dtype=torch.bfloat16)
    bits = torch.zeros(N, H, W, dtype=torch.int32, device="cuda")
    w = torch.randn(32, cin, 3, 3, device="cuda") * 0.05
    wp = K.conv_pack(w, False, cin, math=K.MATH_BF16)
    b = torch.zeros(32, device="cuda")
    for _ in range(3):
        stamps.zero_()
        K.conv_forward(K.Sl(x, cin, 0), wp, b if bits_mode == 1 else None, K.Sl(out, 32, 0), 3, relu=bits_mode == 1,
                       math=K.MATH_BF16, bits=bits, bits_mode=bits_mode)
    torch.cuda.synchronize()
    full = stamps.cpu().numpy().reshape(32, 16)
    full = full[full[:, 0] > 0]
    t = full[:, :9]
    d = np.diff(t, axis=1)
    print(f"cin{cin} bits_mode {bits_mode}: median cycles over {len(d)} waves of 4 tiles:  " +
          "  ".join(f"{n} {int(m)}" for n, m in zip(names, np.median(d, axis=0))) + f"  | tile total {int(np.median(d.sum(1)))}")
    if (full[:, 9] > 0).all():                               # finer stamps inside conv_epilogue (9 entry, 10 loads issued, 11 bias here)
        e = full[:, [5, 9, 10, 11, 6]]
        print("    inside the epilogue: " + "  ".join(f"{n} {int(m)}" for n, m in zip(
            ["to conv_epilogue", "operand loads issued", "bias arrived", "arithmetic + LDS stage"], np.median(np.diff(e, axis=1), axis=0))))
