"""The drop-in surface SURVEY.md section 8b asks for: module / function / argument / parameter names of the
reference's (absent) torch-ngp submodule, checked by introspection on the CPU (no kernels run)."""
import inspect
import math

import torch


def test_extension_level_names():
    from instance_nerf_amd import activation, encoding, gridencoder, raymarching, shencoder
    for name in ("near_far_from_aabb", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
                 "composite_rays_train", "march_rays", "composite_rays"):
        assert callable(getattr(raymarching, name)), name
    sig = inspect.signature(raymarching.march_rays_train)
    assert list(sig.parameters)[:15] == ["rays_o", "rays_d", "bound", "density_bitfield", "C", "H", "nears", "fars",
                                          "step_counter", "mean_count", "perturb", "align", "force_all_rays",
                                          "dt_gamma", "max_steps"]
    assert list(inspect.signature(raymarching.composite_rays_train).parameters)[:5] == \
        ["sigmas", "rgbs", "deltas", "rays", "T_thresh"]
    assert list(inspect.signature(raymarching.near_far_from_aabb).parameters)[:4] == ["rays_o", "rays_d", "aabb", "min_near"]
    assert callable(activation.trunc_exp) and callable(encoding.get_encoder)
    enc = gridencoder.GridEncoder(desired_resolution=2048)
    assert enc.embeddings.shape == (6119864, 2) and enc.output_dim == 32 and enc.offsets.shape == (17,)
    assert shencoder.SHEncoder(degree=4).output_dim == 16
    e, dim = encoding.get_encoder("hashgrid", desired_resolution=2048)
    assert dim == 32 and isinstance(e, gridencoder.GridEncoder)
    e, dim = encoding.get_encoder("sphere_harmonics")
    assert dim == 16


def test_network_constructor_methods_and_state_dict_keys():
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.renderer import NeRFRenderer
    net = NeRFNetwork(encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                      num_layers_color=3, hidden_dim_color=64, bound=1, cuda_ray=True, density_scale=1, min_near=0.2,
                      density_thresh=10, bg_radius=-1)
    assert isinstance(net, NeRFRenderer) and isinstance(net, torch.nn.Module)
    r = inspect.signature(net.render).parameters
    assert list(r)[:4] == ["rays_o", "rays_d", "staged", "max_ray_batch"] and r["max_ray_batch"].default == 4096
    rc = inspect.signature(net.run_cuda).parameters
    for k, v in dict(dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4).items():
        assert rc[k].default == v, k
    assert inspect.signature(net.update_extra_state).parameters["decay"].default == 0.95
    assert inspect.signature(net.update_extra_state).parameters["S"].default == 128
    assert inspect.signature(net.mark_untrained_grid).parameters["S"].default == 64
    for m in ("forward", "density", "color", "get_params", "reset_extra_state", "mark_untrained_grid"):
        assert callable(getattr(net, m)), m
    keys = set(net.state_dict())
    assert {"encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight",
            "color_net.1.weight", "color_net.2.weight", "density_grid", "density_bitfield", "step_counter",
            "aabb_train", "aabb_infer"} <= keys
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert shapes["sigma_net.0.weight"] == (64, 32) and shapes["sigma_net.1.weight"] == (16, 64)
    assert shapes["color_net.0.weight"] == (64, 31) and shapes["color_net.2.weight"] == (3, 64)
    assert shapes["density_grid"] == (1, 128 ** 3) and shapes["density_bitfield"] == (128 ** 3 // 8,)
    groups = net.get_params(1e-2)
    assert all({"params", "lr"} <= set(g) for g in groups) and sum(len(list(g["params"])) for g in groups) == 6
    inst = NeRFNetwork(num_instances=64)
    assert {"instance_encoder.embeddings", "instance_net.0.weight", "instance_net.2.weight"} <= set(inst.state_dict())
    assert tuple(inst.state_dict()["instance_net.2.weight"].shape) == (64, 64)
    inst.freeze_nerf()
    assert not inst.encoder.embeddings.requires_grad and inst.instance_encoder.embeddings.requires_grad


def test_trainer_surface():
    from instance_nerf_amd.nerf.utils import FusedAdam, Trainer, get_rays
    for m in ("train_step", "eval_step", "test_step", "train", "evaluate", "save_checkpoint", "load_checkpoint"):
        assert callable(getattr(Trainer, m)), m
    p = inspect.signature(Trainer.__init__).parameters
    assert {"name", "opt", "model", "criterion", "optimizer", "ema_decay", "lr", "local_rank", "world_size", "device",
            "workspace"} <= set(p)
    # upstream's constructor, in upstream's order (main_nerf.py builds the Trainer with these)
    assert list(p)[1:23] == ["name", "opt", "model", "criterion", "optimizer", "ema_decay", "lr_scheduler", "metrics",
                             "local_rank", "world_size", "device", "mute", "fp16", "eval_interval", "max_keep_ckpt",
                             "workspace", "best_mode", "use_loss_as_metric", "report_metric_at_train", "use_checkpoint",
                             "use_tensorboardX", "scheduler_update_every_step"]
    for m in ("train_one_epoch", "evaluate_one_epoch", "test", "log"):
        assert callable(getattr(Trainer, m)), m
    assert list(inspect.signature(get_rays).parameters)[:6] == ["poses", "intrinsics", "H", "W", "N", "error_map"]
    opt = FusedAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-2)
    assert opt.betas == (0.9, 0.99) and opt.eps == 1e-15 and opt.param_groups[0]["lr"] == 1e-2
    from instance_nerf_amd.nerf import utils
    utils.seed_everything(3)
    a = torch.rand(2)
    utils.seed_everything(3)
    assert torch.equal(a, torch.rand(2))
    x = torch.linspace(0, 1, 50)
    assert torch.allclose(utils.srgb_to_linear(utils.linear_to_srgb(x)), x, atol=1e-4)
    m = utils.PSNRMeter()
    m.update(torch.zeros(4, 3), torch.full((4, 3), 0.1))
    assert abs(m.measure() - 20.0) < 1e-4 and "PSNR" in m.report()
    # mIoU runs over the ids present in the TRUTH (ignore label -1 dropped); a few stray pixels of an id that is not in
    # the view at all are not a class with IoU 0 (rounds 1-3 counted them: measure(all_predicted=True))
    truth = torch.tensor([0, 0, 0, 0, 1, 1, 1, 1, -1, -1])
    pred = torch.tensor([0, 0, 0, 5, 1, 1, 1, 0, 3, 3])
    mi = utils.MIoUMeter(8)
    mi.update(pred, truth)
    assert abs(mi.measure() - (3 / 5 + 3 / 4) / 2) < 1e-9
    assert abs(mi.measure(all_predicted=True) - (3 / 5 + 3 / 4 + 0.0) / 3) < 1e-9 and "mIoU" in mi.report()
    # copy_tensors: one fused launch for GPU tensors, plain copies otherwise (CPU tensors here)
    a, b = torch.zeros(5), torch.arange(5.0)
    c, d = torch.zeros(3, dtype=torch.int64), torch.tensor([4, 5, 6])
    utils.copy_tensors([(a, b), (c, d)])
    assert torch.equal(a, b) and torch.equal(c, d)


def test_fused_adam_state_dict_is_the_torch_adam_layout():
    """Regression (round-1 advisor): upstream checkpoints hold a torch.optim.Adam state dict under 'optimizer'.
    FusedAdam emits and accepts that layout (and still reads round 1's private one)."""
    from instance_nerf_amd.nerf.utils import FusedAdam
    gen = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 2, generator=gen)), torch.nn.Parameter(torch.randn(7, generator=gen))]
    ref = torch.optim.Adam([{"params": [ps[0]], "lr": 1e-2}, {"params": [ps[1]], "lr": 3e-3}], betas=(0.9, 0.99), eps=1e-15)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=gen)
        ref.step()
    sd = ref.state_dict()
    mine = FusedAdam([{"params": [ps[0]], "lr": 1.0}, {"params": [ps[1]], "lr": 1.0}])
    mine.load_state_dict(sd)
    assert mine.step_count == 3 and [g["lr"] for g in mine.param_groups] == [1e-2, 3e-3]
    for i, p in enumerate(ps):
        assert torch.equal(mine.state[p]["exp_avg"], sd["state"][i]["exp_avg"]) and torch.equal(mine.state[p]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    assert isinstance(mine, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.LambdaLR(mine, lambda it: 0.1 ** min(it / 100, 1))     # upstream's scheduler accepts it
    assert sched.get_last_lr() == [1e-2, 3e-3]
    out = mine.state_dict()
    assert set(out) == {"state", "param_groups"} and set(out["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    back = torch.optim.Adam([{"params": [ps[0]]}, {"params": [ps[1]]}])
    back.load_state_dict(out)                                     # torch accepts what FusedAdam emits
    assert float(back.state[ps[1]]["step"]) == 3 and back.param_groups[1]["lr"] == 3e-3
    assert torch.equal(back.state[ps[0]]["exp_avg_sq"], sd["state"][0]["exp_avg_sq"])
    old = FusedAdam([{"params": ps}])
    old.load_state_dict({"step": 9, "lrs": [5e-3], "state": {1: (torch.ones(7), torch.full((7,), 2.0))}})
    assert old.step_count == 9 and old.param_groups[0]["lr"] == 5e-3 and float(old.state[ps[1]]["exp_avg_sq"][0]) == 2.0
    import pytest
    with pytest.raises(ValueError):
        FusedAdam([{"params": ps[:1]}]).load_state_dict(sd)      # different parameter set: caller decides (Trainer warns)


def test_load_checkpoint_accepts_upstream_shapes(tmp_path):
    """A bare model state dict, and a full checkpoint whose optimizer state does not fit (warning, as upstream)."""
    import warnings
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.utils import FusedAdam, Trainer
    kw = dict(cuda_ray=True, bound=1, num_instances=16)
    src = NeRFNetwork(**kw)
    with torch.no_grad():
        src.sigma_net[0].weight.fill_(0.25)
        src.density_grid.fill_(3.0)
    bare, full = str(tmp_path / "bare.pth"), str(tmp_path / "full.pth")
    torch.save(src.state_dict(), bare)
    torch.save({"epoch": 7, "global_step": 1234, "stats": {"loss": [1.0], "results": []}, "model": src.state_dict(),
                "mean_count": 4321, "mean_density": 0.5,
                "optimizer": torch.optim.Adam([torch.nn.Parameter(torch.zeros(2))]).state_dict()}, full)
    make = lambda: Trainer("t", None, NeRFNetwork(**kw), stage="instance", device=torch.device("cpu"),
                           optimizer=lambda model: FusedAdam(model.get_params(1e-2)), workspace=None)   # upstream: optimizer(model)
    a = make()
    a.load_checkpoint(bare)
    assert float(a.model.sigma_net[0].weight[0, 0]) == 0.25 and float(a.model.density_grid[0, 0]) == 3.0
    b = make()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b.load_checkpoint(full)
    assert any("optimizer state not restored" in str(x.message) for x in w)
    assert (b.epoch, b.global_step, b.model.mean_count, b.model.mean_density) == (7, 1234, 4321, 0.5)
    assert float(b.model.sigma_net[0].weight[0, 0]) == 0.25


def test_sph_from_ray_lands_on_the_sphere():
    from instance_nerf_amd import raymarching
    g = torch.Generator().manual_seed(0)
    o = torch.rand(100, 3, generator=g) - 0.5
    d = torch.nn.functional.normalize(torch.randn(100, 3, generator=g), dim=-1)
    c = raymarching.sph_from_ray(o, d, 3.0)
    assert c.shape == (100, 2) and (c.abs() <= 1 + 1e-6).all()
    theta, phi = (c[:, 0] + 1) * math.pi / 2, c[:, 1] * math.pi
    p = 3.0 * torch.stack([torch.sin(theta) * torch.cos(phi), torch.cos(theta), torch.sin(theta) * torch.sin(phi)], -1)
    t = ((p - o) * d).sum(-1, keepdim=True)
    assert (t > 0).all() and torch.allclose(o + t * d, p, atol=1e-4)          # on the ray, in front of the origin


def test_ffmlp_surface_and_cpu_forward():
    """upstream's FFMLP(input_dim, output_dim, hidden_dim, num_layers): one flat `weights` parameter, bias-free ReLU
    layers; equals the explicit chain of matmuls (CPU; the GPU path shares HipLinear's kernels)."""
    from instance_nerf_amd.ffmlp import FFMLP
    m = FFMLP(32, 16, 64, 3)
    assert m.weights.shape == (64 * 32 + 64 * 64 + 16 * 64,) and [tuple(w.shape) for w in m.layer_weights()] == [(64, 32), (64, 64), (16, 64)]
    x = torch.randn(7, 5, 32)
    w0, w1, w2 = m.layer_weights()
    ref = torch.relu(torch.relu(x @ w0.t()) @ w1.t()) @ w2.t()
    out = m(x)
    assert out.shape == (7, 5, 16) and torch.allclose(out, ref, atol=1e-5)
    out.sum().backward()
    assert m.weights.grad is not None and m.weights.grad.abs().sum() > 0


def test_install_aliases_makes_the_references_imports_resolve_here():
    """`import raymarching`, `from nerf.network import NeRFNetwork`, `import roi_align` - the imports of the reference's
    submodule and of /root/reference/nerf_rcnn/model/utils.py:18 - resolve to this package after install_aliases()."""
    import importlib
    import subprocess
    import sys
    code = ("import instance_nerf_amd as ina; s = ina.install_aliases(); assert s == [], s\n"
            "import raymarching, gridencoder, shencoder, activation, encoding, roi_align, ffmlp\n"
            "from nerf.network import NeRFNetwork; from nerf.utils import Trainer, get_rays; from nerf.provider import NeRFDataset\n"
            "from roi_align.roi_align import roi_align_3d\n"
            "from gridencoder import GridEncoder; from encoding import get_encoder\n"
            "assert raymarching.__name__ == 'instance_nerf_amd.raymarching' and NeRFNetwork.__module__ == 'instance_nerf_amd.nerf.network'\n"
            "print('ok')")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_hip_path_refuses_cpu_tensors():
    """No CPU fallback behind the extension-level API: host tensors raise before any launch."""
    import pytest
    from instance_nerf_amd import _lib, raymarching
    import os
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    with pytest.raises(RuntimeError, match="GPU tensor"):
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.2)


def test_registered_custom_ops():
    """BASELINE north star: "PyTorch-ROCm custom ops over a thin C-ABI" - the extension-level entry points are
    registered with the dispatcher (torch.ops.inr.*) with schemas, fake implementations (shape inference without a
    GPU: torch.compile / export can trace through them) and autograd formulas."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from instance_nerf_amd import ops
    from instance_nerf_amd.gridencoder import level_table
    for name in ("near_far_from_aabb", "march_rays_train", "composite_rays_train", "grid_encode", "nerf_forward"):
        assert name in ops.OPS and hasattr(torch.ops.inr, name), name
    sch = str(torch.ops.inr.composite_rays_train.default._schema)
    assert sch.startswith("inr::composite_rays_train(Tensor sigmas, Tensor rgbs, Tensor deltas, Tensor rays, float T_thresh)")
    targs = ops.table_args(level_table(desired_resolution=2048))
    assert len(targs[0]) == 17 and len(targs[1]) == 16 and targs[0][-1] == 6119864
    with FakeTensorMode():
        ro = torch.empty(40, 3)
        n, f = torch.ops.inr.near_far_from_aabb(ro, ro, torch.empty(6), 0.2)
        assert n.shape == f.shape == (40,)
        x, d, dl, rays, counter = torch.ops.inr.march_rays_train(ro, ro, 1.0, torch.empty(128 ** 3 // 8, dtype=torch.uint8),
                                                                 1, 128, n, f, None, 0.0, 1024, 4096)
        assert x.shape == (4096, 3) and dl.shape == (4096, 2) and rays.shape == (40, 3) and counter.shape == (2,)
        enc = torch.ops.inr.grid_encode(x, torch.empty(targs[0][-1], 2, requires_grad=True), 1.0, *targs)
        assert enc.shape == (4096, 32) and enc.requires_grad
        ws, depth, image = torch.ops.inr.composite_rays_train(torch.empty(4096, requires_grad=True), torch.empty(4096, 3), dl,
                                                             rays, 1e-4)
        assert ws.shape == depth.shape == (40,) and image.shape == (40, 3) and image.requires_grad


def test_captured_step_buffer_size_holds_across_small_changes():
    """Trainer._graph_capacity: sizes of the captured step's sample buffers - up to GRAPH_ALIGN, kept while the held size
    still fits and is not more than max(2 units, 1/8) too large (every change re-captures the step's graphs)."""
    from instance_nerf_amd.nerf.utils import Trainer
    t = Trainer.__new__(Trainer)
    a = Trainer.GRAPH_ALIGN
    assert t._graph_capacity(1) == a and t._graph_capacity(a) == a
    assert t._graph_capacity(30 * a - 5) == 30 * a
    assert t._graph_capacity(29 * a - 7) == 30 * a               # wandered down a unit: held
    assert t._graph_capacity(30 * a - 9) == 30 * a
    assert t._graph_capacity(30 * a + 1) == 31 * a               # no longer fits: grows at once
    assert t._graph_capacity(28 * a) == 31 * a                   # 3 units under, 1/8 of 28 units = 3.5: held
    assert t._graph_capacity(20 * a) == 20 * a                   # the scene emptied: shrinks
    assert t._graph_capacity(4 * a) == 4 * a
    assert t._graph_capacity(2 * a + 1) == 4 * a                 # small sizes: two units of slack
    assert t._graph_capacity(a) == a


def test_frame_path_choice_and_pruning_switches_without_a_gpu():
    """Host logic added in round 5 that needs no device: which frame path `NeRFNetwork.forward_table` takes
    (`frame_slices`: off / forced / "auto" only for frame-sized calls on a table whose finest level is 8192+), and the
    instance-stage Trainer's `prune_ignored` plumbing (`_skip_labels`)."""
    import torch
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.utils import Trainer
    net = NeRFNetwork(bound=1, cuda_ray=False, num_instances=4)
    assert net.frame_slices == "auto"
    assert net._use_slices(1 << 22) is False                   # finest level 2048: the fused kernel won every measurement
    net.frame_slices = True
    assert net._use_slices(1 << 10) is True and net._use_slices(0) is False
    net.frame_slices = False
    assert net._use_slices(1 << 22) is False
    small = NeRFNetwork(bound=1, cuda_ray=False, encoder_kwargs={"num_levels": 12})
    small.frame_slices = True
    assert small._use_slices(1 << 22) is False                 # the sliced path needs the 16-level table
    big = NeRFNetwork(bound=4, cuda_ray=False)
    assert int(big.encoder.table["resolutions"][-1]) == 8192 and big.frame_slices == "auto"
    assert big._use_slices(1 << 10) is False                   # a batch, not a frame
    data = {"masks": torch.tensor([[0, -1, 2]])}
    for stage, prune, want in (("instance", True, True), ("instance", False, False), ("nerf", True, False)):
        n = NeRFNetwork(bound=1, cuda_ray=False, num_instances=4)
        tr = Trainer("t", None, n, stage=stage, device=torch.device("cpu"), workspace=None, mute=True, prune_ignored=prune)
        assert tr.prune_ignored is prune
        assert (tr._skip_labels(data) is data["masks"]) is want
        assert tr._skip_labels({}) is None
