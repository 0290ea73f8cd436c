import numpy as np


class DiagOpRef:
    """diag(a) as an oracle-side operator (mul! protocol) without forming n x n."""

    def __init__(self, a):
        self.a = np.asarray(a, dtype=float)

    def mul_(self, dest, v, al=None, be=None):
        if al is None:
            dest[:] = self.a * v
        else:
            dest[:] = al * (self.a * v) + be * dest
        return dest

    def adjoint(self):
        return self
