"""Synthetic EEG epochs and pipeline constants for benchmarks and parity tests.

x[e,c,t] = 1e-5 * (0.6*pink + 0.3*sin(2*pi*f_alpha*t/fs + phi) + 0.1*sin(2*pi*60*t/fs) + 0.2*N(0,1)),
pink = detrended cumulative sum of N(0,1) normalised to unit std, f_alpha ~ U(8,12) per epoch.
Works on any torch device (generated on-device in chunks for the benchmark, on CPU for tests).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SEED = 20250620


def synth_epochs(E: int, C: int = 30, T: int = 1500, sfreq: float = 500.0, seed: int = SEED, device="cpu",
                 chunk: int = 4096, out: torch.Tensor = None) -> torch.Tensor:
    """``out`` ([E, C, T] float32 on ``device``) is filled in place when given (large resident benchmark inputs)."""
    g = torch.Generator(device=device).manual_seed(seed)
    if out is None:
        out = torch.empty((E, C, T), dtype=torch.float32, device=device)
    t = torch.arange(T, dtype=torch.float32, device=device) / sfreq
    ramp = torch.linspace(0.0, 1.0, T, device=device)
    for e0 in range(0, E, chunk):
        n = min(chunk, E - e0)
        w = torch.randn((n, C, T), generator=g, device=device)
        pink = torch.cumsum(w, dim=-1)
        pink = pink - pink[..., :1] - (pink[..., -1:] - pink[..., :1]) * ramp   # remove the linear trend
        pink = pink / (pink.std(dim=-1, keepdim=True) + 1e-12)
        f_alpha = 8.0 + 4.0 * torch.rand((n, 1, 1), generator=g, device=device)
        phi = 2.0 * math.pi * torch.rand((n, C, 1), generator=g, device=device)
        alpha = torch.sin(2.0 * math.pi * f_alpha * t + phi)
        line = torch.sin(2.0 * math.pi * 60.0 * t)
        noise = torch.randn((n, C, T), generator=g, device=device)
        out[e0:e0 + n] = 1e-5 * (0.6 * pink + 0.3 * alpha + 0.1 * line + 0.2 * noise)
    return out


def synth_ica(C: int = 30, seed: int = 97, exclude=(0, 3), pre_whitener: float = 1e-5):
    """Seeded orthogonal unmixing (U), mixing = U^T, identity PCA, zero mean -> kwargs of ica_projection."""
    rs = np.random.RandomState(seed)
    q, _ = np.linalg.qr(rs.randn(C, C))
    return dict(unmixing=q, mixing=q.T, pca_components=np.eye(C), pca_mean=np.zeros(C),
                pre_whitener=pre_whitener, n_components=C, exclude=tuple(exclude))


def synth_codebook(K: int, D: int = 256, seed: int = 7) -> np.ndarray:
    return np.random.RandomState(seed).randn(K, D).astype(np.float32)


def synth_encoder(C: int, D: int = 256, seed: int = 11) -> dict:
    """Kaiming-uniform-like seeded encoder weights with the reference's layer shapes (training/model.py:169-177)."""
    rs = np.random.RandomState(seed)
    shapes = {0: (64, C, 4), 2: (128, 64, 4), 4: (256, 128, 3), 6: (D, 256, 1)}
    w = {}
    for i, shp in shapes.items():
        bound = 1.0 / math.sqrt(shp[1] * shp[2])
        w[f"encoder.{i}.weight"] = rs.uniform(-bound, bound, size=shp).astype(np.float32)
        w[f"encoder.{i}.bias"] = rs.uniform(-bound, bound, size=(shp[0],)).astype(np.float32)
    return w
