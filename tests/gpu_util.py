"""helpers shared by the -m gpu tests (device buffers are torch tensors holding column-major data)."""
import numpy as np
import torch


def dev(A):
    """numpy (m x n) -> torch cuda tensor of shape (n, m): the column-major image of A."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(A, dtype=np.float64).T)).cuda()


def host(t):
    """inverse of dev()."""
    torch.cuda.synchronize()
    return np.asfortranarray(t.detach().cpu().numpy().T)


def zeros(m, n):
    return torch.zeros((n, m), dtype=torch.float64, device="cuda")


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
