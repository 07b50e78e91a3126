"""ICA fit (FastICA) and EOG component scoring on the GPU -- SURVEY.md 8(f)-3.

Mirrors ``ICA(n_components, method='fastica', random_state=97, max_iter='auto').fit(raw, picks='eeg')``
(training/preprocess.py:91-98, multi-train/withpsd.py:127-130, multi-train/CLEAN.py:61,
preprocessing_unsupervised.py:39-40) and ``ica.find_bads_eog(raw, ch_name, threshold=3.0)``
(training/preprocess.py:108, multi-train/withpsd.py:136-139).

What MNE 1.5 does in ``ICA._fit`` (restated from its published algorithm; MNE itself is not installable here,
so this is *parity unpinned at the MNE boundary* like the other MNE-shaped stages -- the FastICA loop itself is
checked against sklearn's ``FastICA``, the back-end MNE calls):
  1. pre-whitener = std over all EEG channels and samples (one scalar), data /= pre-whitener;
  2. PCA with whitening: centre, SVD, sign convention ``svd_flip`` (largest |time course| entry positive),
     explained_variance = S^2 / (T - 1);
  3. n_components: None -> smallest count whose cumulative explained-variance ratio exceeds 0.999999,
     float -> same rule with that fraction, int -> as given;
  4. sklearn ``FastICA(whiten=False, random_state)``: parallel algorithm, logcosh, tol 1e-4,
     ``w_init = RandomState(seed).normal(size=(n, n))``, symmetric decorrelation each iteration;
  5. unmixing /= sqrt(explained_variance[:n]) (columns), mixing = pinv(unmixing), components sorted by
     explained variance (``sum(mixing**2, 0) * sum(sources**2, 1) / (n * T - 1)``, descending).

Every pass over the ``[C, T]`` recording runs on the GPU (``lbm_ica_stats_f32``: second moments for step 1-2,
the fixed-point statistics of step 4; ``lbm_affine_ch_f32``: whitening and projections); the ``n x n`` algebra
between passes (eigh, decorrelation, convergence test) is float64 NumPy on the host.  With a process group the
recording may be sharded over time across ranks: the per-pass sums are all-reduced (RCCL), nothing else moves.
"""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np
import torch

from . import _lib
from .ica import affine_channels, ica_sources_matrix

FUN_IDS = {"logcosh": 0, "exp": 1, "cube": 2, "identity": 3}


def _all_reduce(t: torch.Tensor, group) -> torch.Tensor:
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


@_lib.on_tensor_device
def ica_stats(z: torch.Tensor, W, fun: str = "logcosh", alpha: float = 1.0, T: Optional[int] = None, group=None):
    """One pass: ``gxt[i, j] = sum_t fun(y_i) z_j``, ``gsum[i] = sum_t fun'(y_i)``, ``y = W z`` (float64 sums).

    ``z``: device float32 ``[n, >=T]`` (a row pitch larger than ``T`` is fine).  Returns NumPy float64 arrays.
    """
    if not (isinstance(z, torch.Tensor) and z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and z.stride(1) == 1):
        raise ValueError("ica_stats: z must be a 2-D float32 CUDA tensor with unit inner stride")
    n = z.shape[0]
    T = z.shape[1] if T is None else int(T)
    lib = _lib.load()
    Wd = torch.as_tensor(np.asarray(W, dtype=np.float64)).to(device=z.device, dtype=torch.float32).contiguous()
    if Wd.shape != (n, n):
        raise ValueError(f"ica_stats: W has shape {tuple(Wd.shape)}, expected ({n}, {n})")
    out = torch.empty(n * n + n, dtype=torch.float64, device=z.device)
    nbytes = lib.lbm_ica_stats_workspace_bytes(T, n)
    if nbytes == 0:
        raise ValueError(f"ica_stats: unsupported shape n={n}, T={T} (n <= 64)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    _lib.check(lib.lbm_ica_stats_f32(_lib.ptr(z), z.stride(0), T, n, _lib.ptr(Wd), FUN_IDS[fun], float(alpha), _lib.ptr(out),
                                     _lib.ptr(out[n * n:]), _lib.ptr(ws), _lib.stream_ptr()), "lbm_ica_stats_f32")
    out = _all_reduce(out, group).cpu().numpy()
    return out[: n * n].reshape(n, n), out[n * n:]


@_lib.on_tensor_device
def channel_moments(x: torch.Tensor, group=None):
    """(sum_t x_c, sum_t x_c x_d) in float64 for a device ``[C, T]`` recording (identity pass, W = I)."""
    C = x.shape[0]
    s2, s1 = ica_stats(x, np.eye(C), fun="identity", group=group)
    return s1, 0.5 * (s2 + s2.T)


def _sym_decorrelation(W: np.ndarray) -> np.ndarray:
    """W <- (W W^T)^(-1/2) W   (sklearn.decomposition._fastica._sym_decorrelation)."""
    s, u = np.linalg.eigh(W @ W.T)
    s = np.clip(s, a_min=np.finfo(W.dtype).tiny, a_max=None)
    return np.linalg.multi_dot([u * (1.0 / np.sqrt(s)), u.T, W])


@_lib.on_tensor_device
def fastica_parallel(z: torch.Tensor, w_init, *, fun: str = "logcosh", alpha: float = 1.0, tol: float = 1e-4,
                     max_iter: int = 1000, T_total: Optional[int] = None, group=None):
    """sklearn ``_ica_par`` with the data pass on the GPU.  ``z``: whitened ``[n, T]`` device tensor.

    Returns ``(W [n, n] float64, n_iter)``.  ``T_total``: sample count over all ranks when ``group`` is given.
    """
    p_ = float(z.shape[1] if T_total is None else T_total)
    W = _sym_decorrelation(np.asarray(w_init, dtype=np.float64))
    lim = np.inf
    for ii in range(max_iter):
        gxt, gsum = ica_stats(z, W, fun=fun, alpha=alpha, group=group)
        W1 = _sym_decorrelation(gxt / p_ - (gsum / p_)[:, None] * W)
        lim = float(np.max(np.abs(np.abs(np.einsum("ij,ij->i", W1, W)) - 1.0)))
        W = W1
        if lim < tol:
            break
    else:
        warnings.warn("FastICA did not converge. Consider increasing tolerance or the maximum number of iterations.")
    return W, ii + 1


def _exp_var_ncomp(var: np.ndarray, n: float):
    cvar = np.asarray(var, dtype=np.float64).cumsum()
    cvar /= cvar[-1]
    k = min(int((cvar <= n).sum()) + 1, len(cvar))
    return k, float(cvar[k - 1])


@_lib.on_tensor_device
def ica_fit(data, *, n_components=None, method: str = "fastica", random_state=97, max_iter="auto", fit_params=None,
            group=None) -> dict:
    """Fit on a ``[C, T]`` recording (NumPy or device tensor).  Returns the fitted attributes under the names
    ``ica_apply`` / ``ica_get_sources`` take: ``unmixing, mixing, pca_components, pca_mean, pre_whitener,
    n_components`` plus ``pca_explained_variance, n_iter, exclude`` (empty list)."""
    if method != "fastica":
        raise NotImplementedError(f"ica_fit: method {method!r} (the reference scripts use 'fastica')")
    fit_params = dict(fit_params or {})
    fun = fit_params.pop("fun", "logcosh")
    alpha = float((fit_params.pop("fun_args", None) or {}).get("alpha", 1.0))
    tol = float(fit_params.pop("tol", 1e-4))
    if fit_params:
        raise NotImplementedError(f"ica_fit: unsupported fit_params {sorted(fit_params)}")
    if max_iter == "auto":
        max_iter = 1000                                     # MNE: 'auto' -> 1000 for fastica
    x = _lib.dev_f32(data)
    if x.dim() != 2:
        raise ValueError("ica_fit: data must be [channels, samples]")
    C, T_loc = x.shape
    cnt = _all_reduce(torch.tensor([float(T_loc)], dtype=torch.float64, device=x.device), group)
    T = int(cnt.item())

    # 1. pre-whitener: std over all channels x samples (one channel type)
    s1, s2 = channel_moments(x, group=group)
    tot = float(C) * T
    pre = float(np.sqrt(max(np.trace(s2) / tot - (s1.sum() / tot) ** 2, 0.0)))
    if pre == 0.0:
        pre = 1.0
    # 2. PCA of the pre-whitened data from its covariance (= the SVD of the centred data, T - 1 normalisation)
    mean_w = s1 / T / pre
    cov = (s2 / pre ** 2 - T * np.outer(mean_w, mean_w)) / (T - 1)
    ev, vec = np.linalg.eigh(0.5 * (cov + cov.T))
    order = np.argsort(ev)[::-1]
    ev, V = np.maximum(ev[order], 0.0), vec[:, order].T            # rows of V = components
    # svd_flip (u-based): the largest-magnitude entry of every time course is positive
    proj = affine_channels(x, V / pre, -(V @ mean_w))              # [C, T_loc] device
    amax = proj.abs().max(dim=1)
    idx = amax.indices
    peak = torch.stack([amax.values, proj[torch.arange(C, device=x.device), idx].sign()], dim=1).to(torch.float64)
    if group is not None:                                           # the global peak: rank with the largest magnitude wins
        import torch.distributed as dist
        allp = [torch.empty_like(peak) for _ in range(dist.get_world_size(group))]
        dist.all_gather(allp, peak, group=group)
        allp = torch.stack(allp)                                    # [W, C, 2]
        win = allp[:, :, 0].argmax(dim=0)
        peak = allp[win, torch.arange(C, device=x.device)]
    signs = peak[:, 1].cpu().numpy()
    signs[signs == 0] = 1.0
    V = V * signs[:, None]
    del proj
    # 3. number of components
    ratio = ev / ev.sum()
    if n_components is None:
        n = _exp_var_ncomp(ratio, 0.999999)[0]
    elif isinstance(n_components, float):
        n = _exp_var_ncomp(ratio, n_components)[0]
        if n == 1:
            raise RuntimeError("One PCA component captures most of the explained variance; use an integer n_components")
    else:
        n = int(n_components)
    if not 1 <= n <= min(C, 64):
        raise ValueError(f"ica_fit: n_components={n} outside [1, min(C, 64)]")
    # 4. whitened data and the FastICA fixed point
    norms = np.sqrt(ev[:n])
    norms[norms == 0] = 1.0
    B = V[:n] / norms[:, None]                                       # whitening matrix on pre-whitened, centred data
    z = affine_channels(x, B / pre, -(B @ mean_w))                   # [n, T_loc]
    seed = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    w_init = seed.normal(size=(n, n))
    W, n_iter = fastica_parallel(z, w_init, fun=fun, alpha=alpha, tol=tol, max_iter=int(max_iter), T_total=T, group=group)
    # 5. un-whiten, pseudo-inverse, sort by explained variance
    unmixing = W / norms[None, :]
    mixing = np.linalg.pinv(unmixing)
    czz = B @ (cov * (T - 1)) @ B.T                                  # sum_t z z^T
    src_ss = np.einsum("ij,jk,ik->i", W, czz, W)                     # sum_t sources_i^2
    var = (mixing ** 2).sum(axis=0) * src_ss / (n * T - 1)
    srt = np.argsort(var)[::-1]
    return dict(unmixing=unmixing[srt, :], mixing=mixing[:, srt], pca_components=V, pca_mean=mean_w,
                pre_whitener=np.full((C, 1), pre), n_components=n, pca_explained_variance=ev, n_iter=n_iter, exclude=[])


def _find_outliers(X: np.ndarray, threshold: float = 3.0, max_iter: int = 2) -> np.ndarray:
    """Iterative |z-score| > threshold (mne.preprocessing.bads._find_outliers, tail=0)."""
    X = np.asarray(X, dtype=np.float64)
    mask = np.zeros(len(X), dtype=bool)
    for _ in range(max_iter):
        good = X[~mask]
        sd = good.std()
        zs = np.zeros(len(X)) if sd == 0 or good.size == 0 else np.abs((X - good.mean()) / sd)
        local_bad = (zs > threshold) & ~mask
        mask |= local_bad
        if not local_bad.any():
            break
    return np.where(mask)[0]


def _find_bads_ch(ica: dict, data, target, sfreq: float, threshold: float, l_freq: float, h_freq: float):
    """``ICA._find_bads_ch`` with ``measure='zscore'``: Pearson correlation of the band-passed component time courses with the
    band-passed target channel (``score_sources(score_func='pearsonr', l_freq, h_freq)``), iterative z-score outliers above
    ``threshold``, ordered by |score|.  The sources, the band-pass and the moments all run on the GPU (affine map, FIR,
    second-moment pass); the n-vector of scores is host work."""
    from .filter import filter_data
    x = _lib.dev_f32(data)
    A, b = ica_sources_matrix(unmixing=ica["unmixing"], pca_components=ica["pca_components"], pca_mean=ica["pca_mean"],
                              pre_whitener=ica["pre_whitener"], n_components=ica["n_components"])
    n = A.shape[0]
    if n + 1 > 64:
        raise ValueError("component scoring: at most 63 components")
    tgt = _lib.dev_f32(target).reshape(-1)
    if tgt.numel() != x.shape[1]:
        raise ValueError(f"target channel has {tgt.numel()} samples, the recording {x.shape[1]}")
    stack = torch.empty((n + 1, x.shape[1]), dtype=torch.float32, device=x.device)
    stack[:n] = affine_channels(x, A, b)
    stack[n] = tgt
    stack = filter_data(stack, sfreq, l_freq, h_freq)
    s1, s2 = channel_moments(stack)
    T = stack.shape[1]
    cov = s2 / T - np.outer(s1, s1) / T ** 2
    sd = np.sqrt(np.maximum(np.diag(cov), 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        scores = np.nan_to_num(cov[:n, n] / (sd[:n] * sd[n]))
    bad = _find_outliers(scores, threshold=threshold)
    bad = bad[np.argsort(np.abs(scores[bad]))[::-1]]
    return [int(i) for i in bad], scores


@_lib.on_tensor_device
def find_bads_eog(ica: dict, data, eog, sfreq: float, *, threshold: float = 3.0, l_freq: float = 1.0, h_freq: float = 10.0):
    """Components whose 1-10 Hz time course correlates with the (1-10 Hz) EOG signal: Pearson scores, iterative
    z-score outliers above ``threshold``, ordered by |score| (``ICA.find_bads_eog(measure='zscore')``,
    training/preprocess.py:108, multi-train/withpsd.py:136-139).

    ``data``: the ``[C, T]`` recording the ICA was fitted on; ``eog``: ``[T]`` signal of the EOG (or frontal) channel.
    Returns ``(indices list, scores [n])``.
    """
    return _find_bads_ch(ica, data, eog, sfreq, threshold, l_freq, h_freq)


@_lib.on_tensor_device
def find_bads_ecg(ica: dict, data, ecg, sfreq: float, *, method: str = "correlation", threshold="auto", l_freq: float = 8.0,
                  h_freq: float = 16.0, measure: str = "zscore"):
    """``ica.find_bads_ecg(raw, ch_name=ecg, method='correlation', threshold='auto')`` (multi-train/withpsd.py:145, also
    multi-train/preprocessv2.py:161): components whose 8-16 Hz time course correlates with the 8-16 Hz ECG channel; with
    ``threshold='auto'`` and the z-score measure the outlier threshold is 3.0 (MNE 1.5 ``find_bads_ecg``).  MNE's default
    ``method='ctps'`` (cross-trial phase statistics over detected heart beats) is not what the reference calls and is not
    implemented.

    ``ecg``: the ``[T]`` signal of the ECG channel.  Returns ``(indices list, scores [n])``.
    """
    if method != "correlation":
        raise NotImplementedError("find_bads_ecg: only method='correlation' (the reference's call) is implemented")
    if measure != "zscore":
        raise NotImplementedError("find_bads_ecg: only measure='zscore' (MNE's default) is implemented")
    thr = 3.0 if threshold == "auto" else float(threshold)
    return _find_bads_ch(ica, data, ecg, sfreq, thr, l_freq, h_freq)
