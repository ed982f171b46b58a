"""Runs the device fit of the headline surrogate a few times (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd import PolyModel
d = int(os.environ.get('DIM', 64))
su = PolyModel('quadratic', input_size=d, output_size=1)
P = su.n_param
x = np.random.default_rng(3).normal(size=(2 * P, d))
y = -0.5 * np.sum(x**2, 1)
for _ in range(5):
    su.fit(x, y[:, None], logp=y)
torch.cuda.synchronize()
