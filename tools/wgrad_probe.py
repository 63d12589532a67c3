"""Times inr_linear_wgrad alone (events on the current stream) and checks it against torch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instance_nerf_amd import _lib
from instance_nerf_amd._lib import check, ptr, stream_ptr
lib = _lib.load()
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 205000
x = torch.randn(M, 64, device=dev); gy = torch.randn(M, 64, device=dev)
gw = torch.zeros(64, 64, device=dev)
ws = torch.empty(lib.inr_linear_wgrad_workspace_bytes() // 4, device=dev)
for _ in range(5):
    check(lib.inr_linear_wgrad(ptr(x), ptr(gy), M, 64, 64, ptr(gw), ptr(ws), stream_ptr()), "w")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    check(lib.inr_linear_wgrad(ptr(x), ptr(gy), M, 64, 64, ptr(gw), ptr(ws), stream_ptr()), "w")
e1.record(); torch.cuda.synchronize()
gw.zero_()
check(lib.inr_linear_wgrad(ptr(x), ptr(gy), M, 64, 64, ptr(gw), ptr(ws), stream_ptr()), "w")
ref = gy.t() @ x
print(f"M={M} {e0.elapsed_time(e1) / 50 * 1e3:.1f} us/call  rel err {float((gw - ref).abs().max() / ref.abs().max()):.2e}")
