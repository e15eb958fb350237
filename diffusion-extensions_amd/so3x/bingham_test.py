"""The reference's quantitative evaluation (bingham_test.py): MMD between rotations drawn from a Bingham distribution
and rotations drawn by the trained diffusion.  The 20,000-sample reverse chain is one chain-resident kernel launch per
100 steps and the three 20,000^2 pair sums of the MMD are one fused kernel each (no chunking)."""
import argparse
import os
import pickle

import torch

from .bingham_train import covpairs, RotPredict, loc
from .distributions import Bingham
from .diffusion import SO3Diffusion
from .util import MMD, quat_to_rmat, rmat_gaussian_kernel

__all__ = ["calc_step", "SAMPLES", "NET_SAMPLES", "NET_RUNS", "main"]

SAMPLES = 20_000
NET_SAMPLES = 20_000
NET_RUNS = SAMPLES // NET_SAMPLES


def calc_step(acro, cov, step, weights_dir="weights", samples=SAMPLES, net_samples=NET_SAMPLES, timesteps=1000,
              precision="fp32"):
    """bingham_test.py:15-31"""
    device = torch.device("cuda")
    net = RotPredict(out_type="skewvec", precision=precision).to(device)
    net.load_state_dict(torch.load(os.path.join(weights_dir, f"weights_bing_{acro}_{step}.pt"), map_location=device))
    diff = SO3Diffusion(net, timesteps=timesteps, loss_type="skewvec").to(device)
    bing = Bingham(loc=loc.to(device), covariance_matrix=cov.to(device))
    bing_samples = quat_to_rmat(bing.sample((samples,)))
    diff_samples = torch.cat([diff.p_sample_loop((net_samples,)) for _ in range(samples // net_samples)], dim=0)
    return MMD(bing_samples, diff_samples, rmat_gaussian_kernel, chunksize=4_000).item()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Bingham MMD evaluation")
    ap.add_argument("cov", type=str, help="covariance matrix to use", choices=["sur", "scr", "lur", "lcr"])
    ap.add_argument("--step", type=int, default=100_000)
    ap.add_argument("--weights-dir", default="weights")
    args = ap.parse_args(argv)
    cov, = [c for _, a, c in covpairs if a == args.cov]
    results = {args.step: calc_step(args.cov, cov, args.step, args.weights_dir), "count": SAMPLES}
    with open(f"bingham_mmd_{args.cov}.pkl", "wb") as f:
        pickle.dump(results, f)
    print(results)


if __name__ == "__main__":
    main()
