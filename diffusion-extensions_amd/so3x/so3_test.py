"""The reference's sampling scripts without their plots (so3_test.py, so3_lock_test.py): load trained weights, run the
reverse process with the EXTERNAL per-step loop those scripts use (`process.p_sample(R, t)` with a (1,)-shaped t, the
whole trajectory kept), and report how the samples converge onto the two training modes (the z90 analysis at the end of
so3_test.py).  Every step is one launch of the chain-resident kernel; the trajectory stays on the device."""
import argparse

import torch

from .diffusion import SO3Diffusion
from .util import rmat_dist, rmat_to_euler

__all__ = ["BATCH", "sample_trajectory", "mode_distance", "main"]

BATCH = 512


@torch.no_grad()
def sample_trajectory(process: SO3Diffusion, R: torch.Tensor) -> torch.Tensor:
    """so3_test.py:22-31: res[i] = the state BEFORE reverse step i, i = T-1 .. 0 (so res[T-1] is the initial R and the
    final sample is one more step past res[0]); returns (res [T, B, 3, 3], final [B, 3, 3])."""
    T = process.num_timesteps
    res = torch.zeros((T,) + tuple(R.shape), dtype=torch.float32, device=R.device)
    for i in reversed(range(T)):
        res[i] = R
        R = process.p_sample(R, torch.full((1,), i, device=R.device, dtype=torch.long))
    return res, R


def mode_distance(res: torch.Tensor) -> torch.Tensor:
    """so3_test.py:72-81: geodesic angle of every trajectory point to the nearer of the two training modes (rotations by
    +-90 degrees about z; rmat_dist / sqrt(2) = the angle), the mode chosen per sample by where it ends (res[0])."""
    z90 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], device=res.device)
    plus = rmat_dist(res, z90.expand_as(res).contiguous()) * 0.70710678118
    minus = rmat_dist(res, z90.T.expand_as(res).contiguous()) * 0.70710678118
    close_plus = plus[0] < minus[0]
    return torch.where(close_plus[None], plus, minus)


def main(argv=None, wide=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64 if wide else BATCH)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--weights", default="weights/weights_so3_lock.pt" if wide else "weights/weights_so3.pt")
    ap.add_argument("--out", default=None, help="torch.save the trajectory [T, B, 3, 3] here")
    args = ap.parse_args(argv)
    if wide:
        from .so3_lock_train import RotPredict
    else:
        from .so3_train import RotPredict
    device = torch.device("cuda")
    net = RotPredict(out_type="skewvec", precision=args.precision).to(device)
    net.load_state_dict(torch.load(args.weights, map_location=device))
    net.eval()
    process = SO3Diffusion(net, timesteps=args.timesteps, loss_type="skewvec").to(device)
    # initial rotations: the Q factor of a Gaussian matrix, as the reference (which may have det -1, so3_test.py:24)
    R, _ = torch.linalg.qr(torch.randn((args.batch, 3, 3), device=device))
    res, final = sample_trajectory(process, R)
    x, y, z = rmat_to_euler(res)
    d = mode_distance(res) if not wide else None
    summary = {"timesteps": args.timesteps, "batch": args.batch,
               "final_euler_mean_abs": [float(a[0].abs().mean()) for a in (x, y, z)]}
    if d is not None:
        summary["mode_angle_start_mean"] = float(d[-1].mean())
        summary["mode_angle_end_mean"] = float(d[0].mean())
    print(summary)
    if args.out:
        torch.save({"trajectory": res.cpu(), "final": final.cpu()}, args.out)
    return summary


if __name__ == "__main__":
    main()
