"""Mirror of src/d2d/utils.py (reference): angle wrap + the constant wind field."""
import numpy as np


def norm_mpi_pi(v):
    """Wrap to [-pi, pi) (src/d2d/utils.py:7)."""
    return (v + np.pi) % (2 * np.pi) - np.pi


class WindField:
    """Constant wind; same call signatures as src/d2d/utils.py:10-18."""

    def __init__(self, w=[0., 0.]):
        self.w = w

    def sample_num(self, _x, _y, _t):
        return self.w

    def sample_sym(self, _x, _y, _t):
        return self.w
