"""ORACLE helper (test infrastructure): per-tensor gradient checksums that a full-size golden can afford to store instead of the tensor, and that
still see what sum |t| is blind to -- a sign flip, a transposed or permuted block:

    [ sum |t|,  sum t,  <t, r>,  ||t||_2 ]        r = a fixed pseudo-random vector in [-0.5, 0.5), a function of (tensor name, element index)

`r` comes from integer hashing of the LOGICAL (row-major) element index in int64 arithmetic, so the generator (CPU) and the GPU test build the
identical vector without any RNG stream.  The difference of two projections <t - t', r> is a sample of ||t - t'||_2 / sqrt(12): the tests bound it by
tol * ||t_ref||_2 / sqrt(12) * 3 (three sigma), i.e. they test the RELATIVE L2 ERROR of the whole tensor, sign and placement included.
"""
import zlib

import torch


def projection_vector(n, name, device='cpu'):
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    i = torch.arange(n, dtype=torch.int64, device=device)
    h = (i * 2654435761 + seed * 40503 + 12345) & 0xFFFFFFFF
    h = ((h ^ (h >> 15)) * 2246822519) & 0xFFFFFFFF
    h = ((h ^ (h >> 13)) * 3266489917) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return h.to(torch.float64) / 4294967296.0 - 0.5


def checksum4(t, name):
    """-> [sum |t|, sum t, <t, r(name)>, ||t||_2] in float64 over the logical element order (memory layout does not matter)."""
    t = t.detach().reshape(-1).to(torch.float64)
    r = projection_vector(t.numel(), name, t.device)
    return [float(t.abs().sum()), float(t.sum()), float((t * r).sum()), float(t.pow(2).sum().sqrt())]


def relative_errors(got, want):
    """(abs-sum rel. error, signed-sum error / sum|.|, projection error in units of ||ref||_2 / sqrt(12), L2-norm rel. error) of two checksum4 rows"""
    a, s, p, n = want
    scale = max(n / 12 ** 0.5, 1e-300)
    return (abs(got[0] - a) / max(a, 1e-300), abs(got[1] - s) / max(a, 1e-300), abs(got[2] - p) / scale, abs(got[3] - n) / max(n, 1e-300))
