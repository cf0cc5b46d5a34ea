import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidcom2_amd import synth
import vidcom2_amd.vidcom2 as V
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvc2hip_dbg.so"))
dev = torch.device("cuda:0")
x = synth.make(32, 196, 3584, torch.bfloat16, 0).to(dev)
var_T, var32 = V._channel_variance(x)
v = var32.float().contiguous()
D = v.numel()
print("distinct var values:", torch.unique(v).numel(), "of", D)
mask = torch.empty(D, dtype=torch.uint8, device=dev); cols = torch.empty(D, dtype=torch.int32, device=dev)
order = torch.empty(D, dtype=torch.int32, device=dev); opos = torch.empty(D, dtype=torch.int32, device=dev); spos = torch.empty(D, dtype=torch.int32, device=dev)
t = (ctypes.c_ulonglong * 512)(); vv = (ctypes.c_int * 512)(); n = ctypes.c_int(0)
for with_order in (0, 1):
    for rep in range(3):
        L.vc2_debug_read(t, vv, ctypes.byref(n), 1)
        L.vc2_chan_select(ctypes.c_void_p(v.data_ptr()), ctypes.c_int64(D), ctypes.c_int64(D // 2), ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(cols.data_ptr()),
                          ctypes.c_void_p(order.data_ptr() if with_order else 0), ctypes.c_void_p(opos.data_ptr() if with_order else 0), ctypes.c_void_p(spos.data_ptr() if with_order else 0), ctypes.c_void_p(0))
        L.vc2_debug_read(t, vv, ctypes.byref(n), 0)
    print(f"order={with_order}: {n.value} stamps, total {t[n.value-1]-t[0]} cycles")
    for i in range(1, n.value):
        print(f"   tag {vv[i-1]:>8} -> {vv[i]:>8}: {t[i]-t[i-1]:>7} cycles")
