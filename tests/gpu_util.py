"""helpers shared by the -m gpu tests (device buffers are torch tensors holding column-major data)."""
import numpy as np
import torch


def dev(A):
    """numpy (m x n) -> torch cuda tensor of shape (n, m): the column-major image of A."""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(A, dtype=np.float64).T)).cuda()
    torch.cuda.synchronize()      # torch's stream and the plan's (non-blocking) streams are not ordered
    return t


def host(t):
    """inverse of dev()."""
    torch.cuda.synchronize()
    return np.asfortranarray(t.detach().cpu().numpy().T)


def zeros(m, n):
    t = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()      # the fill runs on torch's stream: finish it before a plan stream writes t
    return t


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
