import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def room():
    from instance_nerf_amd.scene import RoomScene
    return RoomScene()


@pytest.fixture(scope="session")
def room_bitfield(room):
    return room.density_bitfield(128, 1.0)


@pytest.fixture(scope="session")
def level_table():
    from oracle.hashgrid import level_table
    return level_table()


@pytest.fixture(scope="session")
def params_k16(level_table):
    """O(1)-output parity parameters (table U(-1,1)), K = 16 instance logits."""
    from oracle.field import init_params
    return init_params(seed=0, table=level_table, table_std=1.0, K=16)


def scene_rays(room, n=256, cam=0, seed=2):
    import numpy as np
    from oracle.rays import get_rays
    poses, intr, H, W = room.cameras()
    r = get_rays(poses[cam:cam + 1], intr, H, W, N=n, rng=np.random.default_rng(seed))
    return r["rays_o"][0], r["rays_d"][0]


def config0_workload():
    """BASELINE configs[0] as this repository restates it (SURVEY.md section 8d "Config restatement" #1): 100
    Blender-style poses on a sphere of radius 4.0311 (``camera_angle_x`` = 0.6911), brought into the renderer's frame
    the way upstream's loader does (``nerf_matrix_to_ngp``, scale 0.33), 64x64 images, 1024 rays per training step.
    The scene is the synthetic room seen from outside (no dataset exists offline); the supervision is a direction-
    coded colour (|d|), enough for the plumbing the config is about.  Returns a dict of numpy arrays / numbers."""
    import numpy as np
    from instance_nerf_amd.nerf.provider import nerf_matrix_to_ngp
    from instance_nerf_amd.scene import blender_poses
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911)
    blender = blender_poses(100, 4.0311, seed=0)
    # blender_poses builds OpenCV-style look-at matrices (x right, y down, z forward); a Blender transform_matrix has
    # y up and z backward, which is what nerf_matrix_to_ngp expects
    blender[:, :3, 1] *= -1
    blender[:, :3, 2] *= -1
    poses = np.stack([nerf_matrix_to_ngp(p, scale=0.33) for p in blender]).astype(np.float32)
    rng = np.random.default_rng(40)
    steps = [(int(rng.integers(0, 100)), rng.integers(0, H * W, size=1024)) for _ in range(6)]
    return {"H": H, "W": W, "intrinsics": (float(focal), float(focal), W / 2.0, H / 2.0), "poses": poses,
            "steps": steps, "min_near": 0.2, "view": 0}
