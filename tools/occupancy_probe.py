import os, sys, ctypes
# the diagnostic build (NVQ_DEBUG_TOOLS=1 bash build.sh): the shipped libnvq.so has no nvq_debug_* entry points
os.environ.setdefault("NVQ_LIB", os.path.join("continual-learning-for-dynamic-video-quality-enhancement_amd", "libnvq_debug.so"))
sys.path.insert(0, "continual-learning-for-dynamic-video-quality-enhancement_amd")
import torch
from nerve_cl import _nvq as K
torch.zeros(1, device="cuda")
out = (ctypes.c_int * 6)()
K.lib().nvq_debug_conv_occupancy(ctypes.cast(out, ctypes.c_void_p))
print("occupancy conv<2,3,8> conv<2,3,8,cs2> conv<4,3,4> rdb_tail wgrad<3,64> conv<2,3,4>:", list(out))
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "shared/block", getattr(p, "shared_memory_per_block", None), "shared/SM", getattr(p, "shared_memory_per_multiprocessor", None))
