"""so3_lock_test.py of the reference without its mayavi rendering: the same external sampling loop with the 255-wide
residual network (so3_lock_train.RotPredict); see so3_test.py."""
from .so3_test import sample_trajectory, main as _main

__all__ = ["BATCH", "sample_trajectory", "main"]

BATCH = 64


def main(argv=None):
    return _main(argv, wide=True)


if __name__ == "__main__":
    main()
