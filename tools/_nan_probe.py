import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from instance_nerf_amd.nerf import NeRFNetwork, network
from instance_nerf_amd.nerf.provider import NeRFDataset
from instance_nerf_amd.nerf.utils import Trainer
from instance_nerf_amd.scene import RoomScene
dev = torch.device("cuda", 0)
room = RoomScene()
d = tempfile.mkdtemp()
sc = room.write_dataset(d, n_views=32, H=200, W=200, num_instances=16, ignore_frac=0.1)
for trial in range(3):
    torch.manual_seed(trial)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=16).to(dev)
    ds = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, seed=trial)
    tr = Trainer("n", None, net, stage="nerf", device=dev, lr=1e-2, iters=1500, workspace=None, mute=True)
    it = iter(())
    for s in range(400):
        try: b = next(it)
        except StopIteration:
            it = iter(ds); b = next(it)
        l = tr.train_one_step(b)
    print("trial", trial, "nerf loss", float(l), "headroom", network.FX_HEADROOM)
    ds2 = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, mask_dir=sc["mask_dir"], num_instances=16, seed=trial)
    net.mean_density = net.mean_density
    ti = Trainer("i", None, net, stage="instance", device=dev, lr=1e-2, iters=1500, update_extra_interval=10 ** 9, workspace=None, mute=True)
    ti.global_step = 1
    it = iter(())
    tab = net.instance_encoder.embeddings
    for s in range(60):
        try: b = next(it)
        except StopIteration:
            it = iter(ds2); b = next(it)
        l = float(ti.train_one_step(b))
        h = tab._fx_state[:96].cpu().numpy()
        bad = not np.isfinite(l) or not bool(torch.isfinite(tab).all())
        if bad or s < 3:
            print(f"  step {s}: loss {l:.4f} table finite {bool(torch.isfinite(tab).all())} scales0 {int((h[:16]==0).sum())} flags {h[64:80].sum()} near {h[49]} peak {h[80:96].max():.3f} max {h[32:48].max():.2e} ref {h[16:32].max():.2e}")
        if bad:
            g = tab.grad
            print("  grad finite", bool(torch.isfinite(g).all()) if g is not None else None, "labels", b["masks"].min().item(), b["masks"].max().item())
            break
