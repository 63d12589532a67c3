import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import hashgrid, field
from tests.test_gpu_parity import _network, _t
g = np.load("tests/golden/field.npz")
tb = hashgrid.level_table()
p = field.init_params(seed=0, table=tb, table_std=1.0, K=16)
net = _network(p).eval()
with torch.no_grad():
    sigma, rgb = net(_t(g["x"]), _t(g["d"]))
    den = net.density(_t(g["x"]))
    logits = net.instance(_t(g["x"]))
print("sigma rel", np.abs(sigma.cpu().numpy() / g["sigma"] - 1).max())
print("rgb abs", np.abs(rgb.cpu().numpy() - g["rgb"]).max())
print("geo abs", np.abs(den["geo_feat"].cpu().numpy() - g["geo"]).max(), "geo scale", np.abs(g["geo"]).max())
print("logits abs", np.abs(logits.cpu().numpy() - g["logits"]).max(), np.abs(g["logits"]).max())
