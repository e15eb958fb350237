"""Training on Bingham-distributed rotations (reference bingham_train.py): the four covariance settings, the RotPredict
score network (the same class as so3_train's, bingham_train.py:9-46) and the training loop, on the fused kernels."""
import argparse
import os

import torch

from .distributions import Bingham
from .so3_train import RotPredict
from .util import quat_to_rmat

__all__ = ["RotPredict", "BATCH", "loc", "covpairs", "main"]

BATCH = 64
# Small, uncorrelated rotations
cov1 = torch.diag(torch.tensor([1000.0, 0.1, 0.1, 0.1]))
# Small, similar-axis rotations: the ijk parts are correlated.  (bingham_train.py:58-64 lists a fifth row by mistake, which
# MultivariateNormal rejects; this is the 4 x 4 matrix of distributions.py:138-143.)
cov2 = torch.tensor([[1e05, 0.00, 0.00, 0.00],
                     [0.00, 1.00, 0.99, 0.99],
                     [0.00, 0.99, 1.00, 0.99],
                     [0.00, 0.99, 0.99, 1.00]])
# Big, similar-axis rotations
cov3 = torch.tensor([[1.00, 0.00, 0.00, 0.00],
                     [0.00, 1.00, 0.90, 0.90],
                     [0.00, 0.90, 1.00, 0.90],
                     [0.00, 0.90, 0.90, 1.00]])
# unit Gaussian: uniform rotations
cov4 = torch.eye(4)
loc = torch.zeros(4)
covpairs = (("Small Uncorrelated Rotations", "sur", cov1),
            ("Small Correlated Rotations", "scr", cov2),
            ("Large Correlated Rotations", "lcr", cov3),
            ("Large Uncorrelated Rotations", "lur", cov4))


def main(argv=None):
    """bingham_train.py:79-97: for every covariance setting, train a fresh RotPredict on fresh Bingham batches."""
    from .diffusion import SO3Diffusion
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--save-every", type=int, default=1000)
    ap.add_argument("--weights-dir", default="weights")
    ap.add_argument("--cov", nargs="*", default=[a for _, a, _ in covpairs], choices=[a for _, a, _ in covpairs])
    args = ap.parse_args(argv)
    device = torch.device("cuda")
    os.makedirs(args.weights_dir, exist_ok=True)
    for title, acro, cov in covpairs:
        if acro not in args.cov:
            continue
        net = RotPredict(out_type="skewvec", precision=args.precision).to(device)
        net.train()
        process = SO3Diffusion(net, timesteps=args.timesteps, loss_type="skewvec").to(device)
        optim = torch.optim.Adam(process.denoise_fn.parameters(), lr=args.lr, fused=True)
        dist = Bingham(loc.to(device), covariance_matrix=cov.to(device))
        for i in range(args.steps + 1):
            truepos = quat_to_rmat(dist.sample((args.batch,)))
            loss = process(truepos)
            optim.zero_grad()
            loss.backward()
            optim.step()
            if i % 1000 == 0:
                print(title, i, float(loss.detach()), flush=True)
            if i % args.save_every == 0:
                torch.save(net.state_dict(), os.path.join(args.weights_dir, f"weights_bing_{acro}_{i}.pt"))


if __name__ == "__main__":
    main()
