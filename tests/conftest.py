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
