"""Kernel time of a training step on a TRAINED scene (2.4x the samples per step of the bench scene: the occupancy grid
is learned, rays cross whole objects).  Two phases so that the profile holds steady-state steps only:

  python tools/trained_step_profile.py train /tmp/room.pt          1500 NeRF steps, state + occupancy saved
  rocprofv3 --kernel-trace --stats -d <dir> -- python tools/trained_step_profile.py nerf /tmp/room.pt       200 steps
  rocprofv3 --kernel-trace --stats -d <dir> -- python tools/trained_step_profile.py instance /tmp/room.pt   200 steps
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd.nerf import NeRFNetwork                                 # noqa: E402
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset              # noqa: E402
from instance_nerf_amd.nerf.utils import Trainer                              # noqa: E402

dev = torch.device("cuda", 0)
phase, path = sys.argv[1], sys.argv[2]
K = 16
torch.manual_seed(0)
if phase == "train":
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev)
    ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096)
    tr = Trainer("room", None, net, stage="nerf", device=dev, lr=1e-2, iters=1500)
    for _ in range(1500):
        tr.train_one_step(ds.batch())
    torch.save({"state": net.state_dict(), "mean_density": net.mean_density, "iter_density": net.iter_density,
                "mean_count": net.mean_count}, path)
    print("saved", path, "mean_count", net.mean_count)
    sys.exit(0)

init = torch.load(path, map_location=dev)
inst = phase == "instance"
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K if inst else 0).to(dev)
net.load_state_dict(init["state"], strict=False)
net.mean_density, net.iter_density, net.mean_count = init["mean_density"], init["iter_density"], init["mean_count"]
ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096, num_instances=K if inst else 0,
                          ignore_frac=0.1 if inst else 0.0)
# lr as at the end of a run: the steps profiled are steady-state steps, not the first ones of a schedule
tr = Trainer("room_" + phase, None, net, stage=phase, device=dev, lr=1e-3, iters=10 ** 6,
             **(dict(update_extra_interval=10 ** 9) if inst else {}))
tr.global_step = 1
n = int(os.environ.get("N_STEPS", "200"))
batches = [ds.batch() for _ in range(n)]
torch.cuda.synchronize()
t0 = time.perf_counter()
tot = torch.zeros((), dtype=torch.int64, device=dev)
for b in batches:
    tr.train_one_step(b)
    tot += net.last_counter[0]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{phase} stage on the trained room: {dt * 1e3:.3f} ms per step (under the profiler if there is one), "
      f"{int(tot) // n} samples per step, {int(tot) / n / dt / 1e6:.1f} Msamples/s")
