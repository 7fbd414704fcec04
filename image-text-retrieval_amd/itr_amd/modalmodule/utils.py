"""l1norm / l2norm with the reference's signature (itr/modalmodule/utils.py:4-15), on the HIP kernel."""
from .. import ops


def l1norm(X, dim=1, eps=1e-8):
    """X / (sum|X| + eps) along `dim`."""
    return ops.l1norm(X, dim, eps)


def l2norm(X, dim=1, eps=1e-8):
    """X / (sqrt(sum X^2) + eps) along `dim` -- eps AFTER the sqrt."""
    return ops.l2norm(X, dim, eps)
