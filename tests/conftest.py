import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "diffusion-extensions_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: a GPU test that needs ~50 GB of device memory and a minute (still part of -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))

    return G()


def reverse_step_bound(coef, omega_x, omega_x0h, dv, eps32=4e-6):
    """Per-sample bound on max |entry| of (device reverse step - f64 oracle reverse step) for ONE p_sample step
    (reference diffusion.py:291-326) whose network output differs from the oracle's by at most `dv` (absolute, per component;
    0 for an fp32 network, ~2e-2 for bf16 operands).  coef = (a, b, c1, c2) = sqrt_recip_alphas_cumprod[t],
    sqrt_recipm1_alphas_cumprod[t], posterior_mean_coef1[t], posterior_mean_coef2[t]; omega_x / omega_x0h = rotation angles
    of x_t and of the oracle's x0hat (float64 arrays).

        x0hat = so3_scale(x_t, a) @ exp(hat(b v))^T          mean = so3_scale(x0hat, c1) @ so3_scale(x_t, c2)

    * so3_scale(x_t, c2): the matrix log is conditioned like eps32 / (pi - omega_x); times c2 <= 1.
    * x0hat is off by an ANGLE  d = a eps32 / (pi - omega_x)  [fp32 log of x_t scaled by a, up to 20291]  +  b sqrt(3) dv
      [the network's error through the exponential]  +  2e-6 [sine / cosine after range reduction].
    * so3_scale(x0hat, c1) takes log(x0hat) -- conditioned like 1 / (pi - omega_x0h), and the device's x0hat may sit d closer
      to pi than the oracle's -- and multiplies it by c1.  Two log vectors are never further apart than 2 pi, whatever d
      is: at the head of the chain (t >= 998: b > 600) a bf16 network leaves x0hat arbitrary, and what bounds the step is
      c1_t * 2 pi = 1.5e-2 / 1e-2 -- the same saturation the reference's own fp32 arithmetic lives with there.
    """
    import numpy as np
    a, b, c1, c2 = coef
    d = max(a, 1.0) * eps32 / (np.pi - omega_x) + b * np.sqrt(3.0) * dv + 2e-6
    gap = np.maximum(np.pi - omega_x0h - d, 1e-3)
    through_x0hat = c1 * np.minimum(2 * np.pi, d * (1.0 + 1.0 / gap))
    return 2e-5 + eps32 * c2 * (1.0 + 1.0 / (np.pi - omega_x)) + through_x0hat
