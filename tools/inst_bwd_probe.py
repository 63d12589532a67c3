"""Component check of the fused instance-field training kernels against torch (GPU)."""
import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd import _lib
from instance_nerf_amd._lib import check, ptr, stream_ptr
from instance_nerf_amd.nerf import NeRFNetwork

dev = torch.device("cuda")
K = 64
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, num_instances=K).to(dev)
net.instance_encoder.embeddings.data.normal_(0, 0.5)
lib = _lib.load()
f32 = torch.float32
M = 5007
x = torch.rand(M, 3, device=dev) * 2 - 1
w0, w1, w2 = [l.weight.detach() for l in net.instance_net]
pf = torch.empty(lib.inr_instance_packed_floats(K), dtype=f32, device=dev)
pb = torch.empty(lib.inr_instance_bwd_packed_floats(), dtype=f32, device=dev)
check(lib.inr_instance_pack_weights_device(ptr(w0), ptr(w1), ptr(w2), K, ptr(pf), ptr(pb), stream_ptr()), "pack")
logits = torch.empty(M, K, device=dev); enc = torch.empty(M, 32, device=dev)
h1 = torch.empty(M, 64, device=dev); h2 = torch.empty(M, 64, device=dev)
check(lib.inr_instance_forward_train(ptr(x), M, 1.0, ptr(net.instance_encoder.embeddings.data), net.instance_encoder.desc,
                                     ptr(pf), K, ptr(logits), ptr(enc), ptr(h1), ptr(h2), stream_ptr()), "fwd")
with torch.no_grad():
    enc_r = net.instance_encoder(x, bound=1.0)
    h1_r = torch.relu(enc_r @ w0.t()); h2_r = torch.relu(h1_r @ w1.t()); lg_r = h2_r @ w2.t()
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("enc", rel(enc, enc_r), "h1", rel(h1, h1_r), "h2", rel(h2, h2_r), "logits", rel(logits, lg_r))
g = torch.randn(M, K, device=dev)
dz2 = torch.empty(M, 64, device=dev); dz1 = torch.empty(M, 64, device=dev); denc = torch.empty(M, 32, device=dev)
check(lib.inr_instance_backward(ptr(g), K, ptr(h1), ptr(h2), M, ptr(pb), ptr(dz2), ptr(dz1), ptr(denc), stream_ptr()), "bwd")
dz2_r = (g @ w2) * (h2 > 0); dz1_r = (dz2_r @ w1) * (h1 > 0); denc_r = dz1_r @ w0
print("dz2", rel(dz2, dz2_r), "dz1", rel(dz1, dz1_r), "denc", rel(denc, denc_r))
