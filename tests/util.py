"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def synth_points(B, N, seed):
    """Same generator as tests/golden/make_golden.py::synth_points (SURVEY.md 8(d) inputs)."""
    g = torch.Generator().manual_seed(seed)
    pts = 0.1 * torch.randn(B, N, 3, generator=g)
    off = torch.rand(B, 1, 3, generator=g) * torch.tensor([0.4, 0.4, 1.0]) + torch.tensor([-0.2, -0.2, 0.5])
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    return (pts + off).contiguous(), obj


def knn_rows_equivalent(D, idx_a, idx_b):
    """Tie-aware comparison of two neighbour lists for the same distance matrix.

    D (n,n) fp32 distances (the pinned definition), idx_* (n,k).  torch.topk leaves the order of
    equal distances undefined, so the reference's list may differ from the (distance, index)
    policy exactly where distances tie.  Returns (exact_rows, multiset_rows): rows whose index
    lists are identical, and rows whose selected DISTANCE multisets are identical bit for bit.
    """
    idx_a = np.asarray(idx_a, dtype=np.int64)
    idx_b = np.asarray(idx_b, dtype=np.int64)
    exact = (idx_a == idx_b).all(axis=1)
    da = np.sort(np.take_along_axis(D, idx_a, axis=1), axis=1)
    db = np.sort(np.take_along_axis(D, idx_b, axis=1), axis=1)
    same = (da.view(np.int32) == db.view(np.int32)).all(axis=1)
    return exact, same


def rows_without_ties(D, k):
    """Rows whose k+2 smallest distances are pairwise distinct (no tie can affect a top-(k+1))."""
    s = np.sort(D, axis=1)[:, : k + 2]
    return (np.diff(s, axis=1) > 0).all(axis=1)
