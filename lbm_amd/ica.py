"""ICA apply / unmix as one per-sample channel affine map over ``lbm_affine_ch_f32``.

Mirrors ``ica.apply(raw, exclude=ica.exclude)`` (training/preprocess.py:144, multi-train/withpsd.py:152,
preprocessing_unsupervised.py:41), ``ICA.get_sources`` and ``raw.set_eeg_reference('average')``
(preprocessing_unsupervised.py:35).  The fitted attributes are folded on the host in float64 into
``A [Cout, Cin]`` and ``b [Cout]``; the GPU applies ``y = A x + b`` to every sample.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib


def _vec(v, C):
    if v is None:
        return np.zeros(C)
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    return np.full(C, float(a[0])) if a.size == 1 else a.reshape(C)


def ica_projection(*, unmixing, mixing, pca_components, pca_mean, pre_whitener, n_components: int,
                   exclude: Sequence[int] = ()):
    """out = s*(P(x/s - mu) + mu)  ->  A = diag(s) P diag(1/s), b = s*(mu - P mu)   (float64)."""
    pca = np.asarray(pca_components, dtype=np.float64)
    n_pca, C = pca.shape
    mu, s = _vec(pca_mean, C), _vec(pre_whitener, C)
    U = np.eye(n_pca)
    U[:n_components, :n_components] = np.asarray(unmixing, dtype=np.float64)
    U = U @ pca
    M = np.eye(n_pca)
    M[:n_components, :n_components] = np.asarray(mixing, dtype=np.float64)
    M = pca.T @ M
    excl = np.asarray(sorted(set(int(i) for i in exclude)), dtype=int)
    keep = np.concatenate([np.setdiff1d(np.arange(n_components), excl), np.arange(n_components, n_pca)]).astype(int)
    P = M[:, keep] @ U[keep, :]
    return (s[:, None] * P) / s[None, :], s * (mu - P @ mu)


def ica_sources_matrix(*, unmixing, pca_components, pca_mean, pre_whitener, n_components: int):
    """S = W (x/s - mu), W = unmixing @ pca_components[:n_comp]  ->  A = W diag(1/s), b = -W mu."""
    pca = np.asarray(pca_components, dtype=np.float64)
    C = pca.shape[1]
    mu, s = _vec(pca_mean, C), _vec(pre_whitener, C)
    W = np.asarray(unmixing, dtype=np.float64) @ pca[:n_components]
    return W / s[None, :], -(W @ mu)


def average_reference_matrix(C: int) -> np.ndarray:
    return np.eye(C) - np.full((C, C), 1.0 / C)


@_lib.on_tensor_device
def affine_channels(data, A, b=None):
    """y[e,o,t] = sum_c A[o,c] x[e,c,t] + b[o] on the GPU.  data: [E, Cin, T] (or [Cin, T])."""
    was_numpy = not isinstance(data, torch.Tensor)
    x = _lib.dev_f32(data)
    squeeze = x.dim() == 2
    if squeeze:
        x = x.unsqueeze(0)
    E, Cin, T = x.shape
    A_t = A if isinstance(A, torch.Tensor) else torch.as_tensor(np.asarray(A, dtype=np.float64))
    A_t = A_t.to(device=x.device, dtype=torch.float32).contiguous()
    Cout = A_t.shape[0]
    if A_t.shape != (Cout, Cin):
        raise ValueError(f"A has shape {tuple(A_t.shape)}, expected ({Cout}, {Cin})")
    b_t = None
    if b is not None:
        b_t = (b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b, dtype=np.float64)))
        b_t = b_t.to(device=x.device, dtype=torch.float32).contiguous()
    y = torch.empty((E, Cout, T), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().lbm_affine_ch_f32(_lib.ptr(x), _lib.ptr(y), E, Cin, Cout, T, _lib.ptr(A_t), _lib.ptr(b_t),
                                             _lib.stream_ptr()), "lbm_affine_ch_f32")
    if squeeze:
        y = y[0]
    return y.cpu().numpy().astype(np.float64) if was_numpy else y


@_lib.on_tensor_device
def ica_apply(data, *, unmixing, mixing, pca_components, pca_mean, pre_whitener, n_components, exclude=()):
    A, b = ica_projection(unmixing=unmixing, mixing=mixing, pca_components=pca_components, pca_mean=pca_mean,
                          pre_whitener=pre_whitener, n_components=n_components, exclude=exclude)
    return affine_channels(data, A, b)


@_lib.on_tensor_device
def ica_get_sources(data, *, unmixing, pca_components, pca_mean, pre_whitener, n_components):
    A, b = ica_sources_matrix(unmixing=unmixing, pca_components=pca_components, pca_mean=pca_mean,
                              pre_whitener=pre_whitener, n_components=n_components)
    return affine_channels(data, A, b)
