"""torch.func plumbing shared by AutoDiffDynamics / AutoDiffCost (reference: jax jacfwd / hessian,
traoptlibrary/traopt_dynamics.py:133-270, traopt_cost.py:113-290).

The user function takes torch tensors (x [n], u [m], i) and returns a tensor; derivatives come from
torch.func.jacfwd / hessian exactly where the reference uses the jax transforms of the same name.  Every
derivative also exists in a knot-batched form (torch.func.vmap over the horizon) so that the Euclidean iLQR
evaluates one rollout's N Jacobians in a single call, on whatever device the tensors live on.
"""
import numpy as np
import torch
from torch.func import hessian, jacfwd, vmap

DTYPE = torch.float64


def as_t(a, device=None):
    if isinstance(a, torch.Tensor):
        return a.to(DTYPE)
    return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=DTYPE, device=device)


def to_np(t):
    return t.detach().cpu().numpy()


class Derivs:
    """f, df/dx, df/du (and the three second derivatives) of fn(x, u, i), single and knot-batched."""

    def __init__(self, fn, second=True, device=None):
        self.device = device
        self.fn = fn
        self.fx = jacfwd(fn, argnums=0)
        self.fu = jacfwd(fn, argnums=1)
        self.second = second
        if second:
            self.fxx = hessian(fn, argnums=0)
            self.fux = jacfwd(jacfwd(fn, argnums=1), argnums=0)
            self.fuu = hessian(fn, argnums=1)

    def one(self, which, x, u, i):
        g = getattr(self, which)
        return to_np(g(as_t(x, self.device), as_t(u, self.device).reshape(-1), i))

    def batch(self, which, xs, us, i0=0):
        """which over knots i0 .. i0+len(xs)-1; the knot index is passed as a tensor element."""
        g = getattr(self, which)
        xs = as_t(xs, self.device); us = as_t(us, self.device)
        idx = torch.arange(i0, i0 + xs.shape[0], device=xs.device)
        return to_np(vmap(g, in_dims=(0, 0, 0))(xs, us, idx))
