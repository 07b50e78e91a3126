"""VQ-VAE tokenizer: drop-in for the quantizer part of the reference's training/model.py.

Same class names, constructor arguments, parameter/buffer names (so ``training/vq_vae_model.pth`` loads
unchanged) and return values as ``VectorQuantizerEMA`` (training/model.py:66-158), ``VQVAE`` (:162-214)
and ``encode_data_for_transformer`` (:411-427); the arithmetic on the encode path runs in
liblbm_hip.so (fp32 MFMA conv + fp32 MFMA distance/argmin), never on a CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .constants import VQ_EMBEDDING_DIM


# ------------------------------------------------------------------------------------------ functional layer
@_lib.on_tensor_device
def rownorm2(m: torch.Tensor) -> torch.Tensor:
    m = _lib.dev_f32(m)
    out = torch.empty((m.shape[0],), dtype=torch.float32, device=m.device)
    _lib.check(_lib.load().lbm_rownorm2_f32(_lib.ptr(m), _lib.ptr(out), m.shape[0], m.shape[1], _lib.stream_ptr()),
               "lbm_rownorm2_f32")
    return out


def fold_projection(codebook, proj_weight, proj_bias=None):
    """Fold the encoder's last 1x1 convolution (training/model.py:176, no activation after it) into the codebook.

    With z = W h + b:  |z - e_k|^2 - |z|^2 = (|e_k|^2 - 2 b.e_k) - 2 h.(W^T e_k), so the tokens of ``VQVAE.encode``
    follow from the third layer's output h with ``codebook' = E W`` and ``e_sqnorm' = |e|^2 - 2 E b`` (float64 on the
    host, rounded once) and ``vq_argmin(h, codebook', e_sqnorm', no_xnorm=True)`` -- the 256x256 projection GEMM
    and the z tensor disappear.  Tokens differ from the unfolded path only at fp32 near-ties.
    Returns ``(codebook' [K, Cin], e_sqnorm' [K])`` as float32 tensors on the codebook's device.
    """
    dev = codebook.device
    E = codebook.detach().to("cpu", torch.float64)                                       # tiny: float64 on the host
    W = proj_weight.detach().to("cpu", torch.float64).reshape(proj_weight.shape[0], -1)  # [D, Cin] (ksize 1)
    if W.shape[0] != E.shape[1]:
        raise ValueError(f"projection has {W.shape[0]} output channels, codebook dim is {E.shape[1]}")
    c = (E * E).sum(dim=1)
    if proj_bias is not None:
        c = c - 2.0 * (E @ proj_bias.detach().to("cpu", torch.float64))
    return (E @ W).to(torch.float32).contiguous().to(dev), c.to(torch.float32).contiguous().to(dev)


def dedup_codebook(codebook: torch.Tensor, e_sqnorm: torch.Tensor, pad_to: int = 64):
    """The sub-codebook a first-minimum search actually needs.

    ``get_code_indices`` (training/model.py:152-158) returns the FIRST minimum, so a row that repeats an earlier row bit for bit --
    value and ``e_sqnorm`` alike -- can never be returned, whatever the input.  (The shipped checkpoint has 100 all-zero rows among its
    512: codes the EMA never used.)  Returns ``None`` if no row repeats, else ``(codebook_c [Kp, D], e_sqnorm_c [Kp], code_map [Kp]
    int64, Kc)``: the first occurrences in ascending order (ties keep resolving to the lowest original index), padded to a multiple of
    ``pad_to`` rows with codes that never win (zero rows, |e|^2 = 3e38); ``code_map[k]`` is searched row k's original number (padding:
    0, never produced).  Search with (codebook_c, e_sqnorm_c), then ``lbm_vq_remap_codes_i64`` -- tokens and histogram come out
    exactly as from the full codebook."""
    cb = codebook.detach().to("cpu", torch.float32).contiguous().numpy()
    ee = e_sqnorm.detach().to("cpu", torch.float32).contiguous().numpy()
    K, D = cb.shape
    key = np.concatenate([cb.view(np.uint32), ee.view(np.uint32)[:, None]], axis=1)
    _, first = np.unique(key, axis=0, return_index=True)
    if first.size == K:
        return None
    first = np.sort(first)
    Kc = int(first.size)
    Kp = -(-Kc // pad_to) * pad_to
    cbc = np.zeros((Kp, D), np.float32)
    eec = np.full((Kp,), 3e38, np.float32)
    cmap = np.zeros((Kp,), np.int64)
    cbc[:Kc], eec[:Kc], cmap[:Kc] = cb[first], ee[first], first
    dev = codebook.device
    return torch.from_numpy(cbc).to(dev), torch.from_numpy(eec).to(dev), torch.from_numpy(cmap).to(dev), Kc


@_lib.on_tensor_device
def remap_codes(idx: torch.Tensor, code_map: torch.Tensor, n_codes: int, hist_searched: torch.Tensor = None,
                hist: torch.Tensor = None) -> torch.Tensor:
    """In place: indices of a searched sub-codebook (``dedup_codebook``) -> the caller's code numbers (``lbm_vq_remap_codes_i64``);
    with the two histograms, the counts over searched rows are added into the full histogram and cleared."""
    if idx.dtype != torch.int64 or not idx.is_contiguous():
        raise ValueError("remap_codes: idx must be a contiguous int64 tensor")
    _lib.check(_lib.load().lbm_vq_remap_codes_i64(_lib.ptr(idx), idx.numel(), _lib.ptr(code_map), int(n_codes), _lib.ptr(hist_searched),
                                                  _lib.ptr(hist), _lib.stream_ptr()), "lbm_vq_remap_codes_i64")
    return idx


@_lib.on_tensor_device
def vq_split_tables(codebook: torch.Tensor) -> torch.Tensor:
    """Device table of the codebook cut into three bf16 pieces per value (``lbm_vq_split_tables_init``), the form
    ``vq_argmin(..., split_tables=...)`` / ``lbm_vq_argmin_bf16x6_f32`` consume."""
    cb = _lib.dev_f32(codebook)
    K, D = cb.shape
    lib = _lib.load()
    nbytes = lib.lbm_vq_split_tables_bytes(K, D)
    if nbytes == 0:
        raise NotImplementedError(f"vq_split_tables: unsupported codebook shape ({K}, {D}) (D <= 256)")
    tab = torch.empty(nbytes, dtype=torch.uint8, device=cb.device)
    _lib.check(lib.lbm_vq_split_tables_init(_lib.ptr(tab), _lib.ptr(cb), K, D, _lib.stream_ptr()), "lbm_vq_split_tables_init")
    return tab


@_lib.on_tensor_device
def vq_argmin(x: torch.Tensor, codebook: torch.Tensor, e_sqnorm: torch.Tensor = None, *, channel_major: bool = False,
              return_min: bool = False, hist: torch.Tensor = None, no_xnorm: bool = False, split_tables: torch.Tensor = None,
              screened: bool = False, return_refined: bool = False):
    """Nearest-code indices (int64).

    ``split_tables`` (from ``vq_split_tables(codebook)``) selects the bf16-piece kernel (fp32-grade products on the
    bf16 matrix cores); without it the fp32-MFMA kernel runs.  ``screened=True`` (needs ``split_tables``, no
    ``return_min``) returns the fp32-MFMA kernel's indices exactly, from a bf16 screening pass plus an fp32 search of
    the rows the screen cannot settle (``lbm_vq_argmin_screened_f32``); ``return_refined`` adds that row count.

    x: [N, D] rows (any row stride that is a multiple of 1 float), or with ``channel_major`` the encoder
    output z[B, D, S] (N = B*S rows, no permute/contiguous copy).  ``hist`` ([K] int64/uint64 counts) is
    ADDED to when given.
    """
    if not x.is_cuda or x.dtype != torch.float32:
        x = _lib.dev_f32(x)
    cb = _lib.dev_f32(codebook)
    K, D = cb.shape
    if e_sqnorm is None:
        e_sqnorm = rownorm2(cb)
    S = 0
    if channel_major:
        x = _pitched_rows(x)
        B, Dx, S = x.shape
        N, ldx, layout = B * S, (x.stride(1) if B * Dx > 1 else S), _lib.VQ_CHANNEL_MAJOR
    else:
        if x.dim() != 2:
            raise ValueError("x must be [N, D]")
        if x.stride(1) != 1:
            x = x.contiguous()
        N, Dx = x.shape
        ldx, layout = (x.stride(0) if N > 1 else Dx), _lib.VQ_ROWS
    if Dx != D:
        raise ValueError(f"input dim {Dx} != codebook dim {D}")
    idx = torch.empty((N,), dtype=torch.int64, device=x.device)
    mind = torch.empty((N,), dtype=torch.float32, device=x.device) if return_min else None
    if hist is not None and (hist.dtype != torch.int64 or hist.numel() != K or not hist.is_contiguous()):
        raise ValueError("hist must be a contiguous int64 tensor of K entries")
    if no_xnorm:
        layout |= _lib.VQ_NO_XNORM
    if screened:
        if split_tables is None or return_min:
            raise ValueError("screened=True needs split_tables and has no minimum-distance output")
        lib = _lib.load()
        ws = torch.empty(lib.lbm_vq_screen_workspace_bytes(N, K), dtype=torch.uint8, device=x.device)
        _lib.check(lib.lbm_vq_argmin_screened_f32(_lib.ptr(x), ldx, layout, S, _lib.ptr(cb), _lib.ptr(split_tables), _lib.ptr(e_sqnorm),
                                                  _lib.ptr(idx), _lib.ptr(hist), _lib.ptr(ws), N, D, K, _lib.stream_ptr()),
                   "lbm_vq_argmin_screened_f32")
        return (idx, int(ws[:4].view(torch.int32).item())) if return_refined else idx
    if split_tables is not None:
        _lib.check(_lib.load().lbm_vq_argmin_bf16x6_f32(_lib.ptr(x), ldx, layout, S, _lib.ptr(split_tables), _lib.ptr(e_sqnorm),
                                                        _lib.ptr(idx), _lib.ptr(mind), _lib.ptr(hist), N, D, K, _lib.stream_ptr()),
                   "lbm_vq_argmin_bf16x6_f32")
        return (idx, mind) if return_min else idx
    _lib.check(_lib.load().lbm_vq_argmin_f32(_lib.ptr(x), ldx, layout, S, _lib.ptr(cb), _lib.ptr(e_sqnorm), _lib.ptr(idx),
                                             _lib.ptr(mind), _lib.ptr(hist), N, D, K, _lib.stream_ptr()), "lbm_vq_argmin_f32")
    return (idx, mind) if return_min else idx


@_lib.on_tensor_device
def vq_gather_loss(x: torch.Tensor, codebook: torch.Tensor, idx: torch.Tensor, want_quantized: bool = True):
    """(quantized with straight-through arithmetic [N, D] or None, sum of squared error as a float64 scalar tensor)."""
    x = _lib.dev_f32(x)
    cb = _lib.dev_f32(codebook)
    N, D = x.shape
    q = torch.empty_like(x) if want_quantized else None
    err = torch.zeros((1,), dtype=torch.float64, device=x.device)
    _lib.check(_lib.load().lbm_vq_gather_loss_f32(_lib.ptr(x), _lib.ptr(cb), _lib.ptr(idx), _lib.ptr(q), _lib.ptr(err), N, D,
                                                  cb.shape[0], _lib.stream_ptr()), "lbm_vq_gather_loss_f32")
    return q, err


@_lib.on_tensor_device
def vq_ema_sums(flat_input: torch.Tensor, idx: torch.Tensor, K: int):
    """(dw [K, D] fp32, counts [K] int64): the segmented sums of training/model.py:109-110,122 (deterministic)."""
    x = _lib.dev_f32(flat_input)
    if x.stride(1) != 1:
        x = x.contiguous()
    N, D = x.shape
    lib = _lib.load()
    nbytes = int(lib.lbm_vq_ema_workspace_bytes(N, D, K))
    if nbytes == 0:
        raise NotImplementedError(f"EMA sums support K <= 8192 (got K={K})")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    dw = torch.empty((K, D), dtype=torch.float32, device=x.device)
    counts = torch.empty((K,), dtype=torch.int64, device=x.device)
    idx = idx.reshape(-1).to(torch.int64).contiguous()
    _lib.check(lib.lbm_vq_ema_sums_f32(_lib.ptr(x), x.stride(0) if N > 1 else D, _lib.ptr(idx), _lib.ptr(dw), _lib.ptr(counts),
                                       _lib.ptr(ws), N, D, K, _lib.stream_ptr()), "lbm_vq_ema_sums_f32")
    return dw, counts


@_lib.on_tensor_device
def vq_ema_apply(ema_cluster_size: torch.Tensor, ema_w: torch.Tensor, weight: torch.Tensor, dw: torch.Tensor,
                 counts: torch.Tensor, decay: float, epsilon: float):
    """In-place EMA codebook update (training/model.py:109-126) on fp32 CUDA tensors."""
    for t in (ema_cluster_size, ema_w, weight, dw):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("EMA state must be contiguous float32 CUDA tensors")
    K, D = ema_w.shape
    _lib.check(_lib.load().lbm_vq_ema_apply_f32(_lib.ptr(ema_cluster_size), _lib.ptr(ema_w), _lib.ptr(weight), _lib.ptr(dw),
                                                _lib.ptr(counts.contiguous()), K, D, float(decay), float(1 - decay),
                                                float(epsilon), _lib.stream_ptr()), "lbm_vq_ema_apply_f32")


@_lib.on_tensor_device
def perplexity_from_hist(hist: torch.Tensor) -> torch.Tensor:
    """exp(-sum p log(p + 1e-10)), p = hist / sum(hist)  (training/model.py:142-144)."""
    out = torch.empty((1,), dtype=torch.float32, device=hist.device)
    _lib.check(_lib.load().lbm_perplexity_u64(_lib.ptr(hist), _lib.ptr(out), hist.numel(), _lib.stream_ptr()),
               "lbm_perplexity_u64")
    return out[0]


def _pitch4(n: int) -> int:
    return (n + 3) & ~3


def _pitched_rows(x: torch.Tensor) -> torch.Tensor:
    """x[B, C, L] as the kernels address it: unit stride in L, one pitch between consecutive (b, c) rows.
    Contiguous tensors and the [..., :L] views of pitched buffers pass through; anything else is copied."""
    B, Cc, L = x.shape
    if x.stride(2) == 1 and (B == 1 or x.stride(0) == Cc * x.stride(1)) and (B * Cc == 1 or x.stride(1) >= L):
        return x
    return x.contiguous()


def pitched_empty(B: int, Cc: int, L: int, device) -> torch.Tensor:
    """[B, C, L] float32 view of a buffer whose rows are padded to a multiple of 4 floats (16-byte rows)."""
    return torch.empty((B, Cc, _pitch4(L)), dtype=torch.float32, device=device)[:, :, :L]


@_lib.on_tensor_device
def conv_split_tables(weight: torch.Tensor) -> torch.Tensor:
    """Device table of conv weights [Cout, Cin, k] cut into three bf16 pieces per value (``lbm_conv1d_split_tables_init``),
    the form ``conv1d(..., split_tables=...)`` / ``lbm_conv1d_bf16x6_f32`` consume."""
    w = _lib.dev_f32(weight)
    Cout, Cin, ks = w.shape
    lib = _lib.load()
    nbytes = lib.lbm_conv1d_split_tables_bytes(Cin, Cout, ks)
    if nbytes == 0:
        raise NotImplementedError(f"conv_split_tables: unsupported weight shape {tuple(w.shape)}")
    tab = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.lbm_conv1d_split_tables_init(_lib.ptr(tab), _lib.ptr(w), Cin, Cout, ks, _lib.stream_ptr()),
               "lbm_conv1d_split_tables_init")
    return tab


@_lib.on_tensor_device
def conv_wino3_tables(weight: torch.Tensor) -> torch.Tensor:
    """Device table of the Winograd-transformed weights of a 3-tap, stride-1, padding-1 layer [Cout, Cin, 3]
    (``lbm_conv1d_wino3_tables_init``), the form ``conv1d(..., wino_tables=...)`` / ``lbm_conv1d_wino3_f32`` consume."""
    w = _lib.dev_f32(weight)
    Cout, Cin, ks = w.shape
    lib = _lib.load()
    nbytes = lib.lbm_conv1d_wino3_tables_bytes(Cin, Cout) if ks == 3 else 0
    if nbytes == 0:
        raise NotImplementedError(f"conv_wino3_tables: unsupported weight shape {tuple(w.shape)} (3 taps, Cin % 8 == 0, Cout >= 64)")
    tab = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.lbm_conv1d_wino3_tables_init(_lib.ptr(tab), _lib.ptr(w), Cin, Cout, _lib.stream_ptr()), "lbm_conv1d_wino3_tables_init")
    return tab


@_lib.on_tensor_device
def conv1d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, stride: int = 1, padding: int = 0,
           relu: bool = False, pitched_out: bool = False, split_tables: torch.Tensor = None,
           wino_tables: torch.Tensor = None) -> torch.Tensor:
    """nn.Conv1d (+ReLU) forward on fp32 MFMA.  x: [B, Cin, L] (contiguous or a pitched view); weight: [Cout, Cin, k].
    ``pitched_out`` returns a [B, Cout, Lout] view of a row-padded buffer (what the next layer reads fastest).
    ``split_tables`` (from ``conv_split_tables(weight)``) selects the bf16-piece kernel (fp32-grade products on the
    bf16 matrix cores; results within fp32 rounding of the default kernel).  ``wino_tables`` (from
    ``conv_wino3_tables(weight)``; 3 taps, stride 1, padding 1) takes the Winograd-transformed weights from that table: the
    default kernel's results bit for bit, a few percent faster."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        x = _lib.dev_f32(x)
    x = _pitched_rows(x)
    w = _lib.dev_f32(weight)
    b = _lib.dev_f32(bias) if bias is not None else None
    B, Cin, Lin = x.shape
    Cout, Cin_w, ks = w.shape
    if Cin_w != Cin:
        raise ValueError(f"weight expects {Cin_w} input channels, got {Cin}")
    Lout = (Lin + 2 * padding - ks) // stride + 1
    y = pitched_empty(B, Cout, Lout, x.device) if pitched_out else torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
    ldx = x.stride(1) if B * Cin > 1 else Lin
    ldy = y.stride(1) if B * Cout > 1 else Lout
    if wino_tables is not None:
        if (ks, stride, padding) != (3, 1, 1):
            raise ValueError("wino_tables is for 3-tap, stride-1, padding-1 layers")
        _lib.check(_lib.load().lbm_conv1d_wino3_f32(_lib.ptr(x), ldx, _lib.ptr(wino_tables), _lib.ptr(b), _lib.ptr(y), ldy, B, Cin, Lin,
                                                    Cout, int(relu), _lib.stream_ptr()), "lbm_conv1d_wino3_f32")
        return y
    if split_tables is not None:
        _lib.check(_lib.load().lbm_conv1d_bf16x6_f32(_lib.ptr(x), ldx, _lib.ptr(split_tables), _lib.ptr(b), _lib.ptr(y), ldy, B, Cin,
                                                     Lin, Cout, ks, stride, padding, int(relu), _lib.stream_ptr()),
                   "lbm_conv1d_bf16x6_f32")
        return y
    _lib.check(_lib.load().lbm_conv1d_pitched_f32(_lib.ptr(x), ldx, _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), ldy, B, Cin, Lin, Cout,
                                                  ks, stride, padding, int(relu), _lib.stream_ptr()), "lbm_conv1d_pitched_f32")
    return y


@_lib.on_tensor_device
def encoder_forward(x: torch.Tensor, enc: "nn.Sequential | dict", pitched_out: bool = False) -> torch.Tensor:
    """VQVAE.encoder (training/model.py:169-177) on the HIP conv kernel: [B, C, L] -> [B, D, L//4].
    Intermediate activations live in row-padded buffers (16-byte rows)."""
    if isinstance(enc, dict):
        g = lambda i, n: enc[f"encoder.{i}.{n}"]  # noqa: E731
    else:
        g = lambda i, n: getattr(enc[i], n)  # noqa: E731
    z = conv1d(x, g(0, "weight"), g(0, "bias"), 2, 1, True, pitched_out=True)
    z = conv1d(z, g(2, "weight"), g(2, "bias"), 2, 1, True, pitched_out=True)
    z = conv1d(z, g(4, "weight"), g(4, "bias"), 1, 1, True, pitched_out=True)
    return conv1d(z, g(6, "weight"), g(6, "bias"), 1, 0, False, pitched_out=pitched_out)


# ------------------------------------------------------------------------------------------ modules
class VectorQuantizerEMA(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay, epsilon=1e-5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.decay = decay
        self.epsilon = epsilon
        self.embedding = nn.Embedding(self.num_embeddings, self.embedding_dim)
        self.embedding.weight.data.uniform_(-1. / self.num_embeddings, 1. / self.num_embeddings)
        self.register_buffer('ema_cluster_size', torch.zeros(num_embeddings))
        self.register_buffer('ema_w', torch.zeros_like(self.embedding.weight.data))
        self._ee_key = None
        self._ee = None
        self._cb_epoch = 0          # counts in-place codebook rewrites (EMA steps), which autograd versions do not see

    def _e_sqnorm(self) -> torch.Tensor:
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._ee_key != key:
            self._ee = rownorm2(w.detach())
            self._ee_key = key
        return self._ee

    def get_code_indices(self, flat_input):
        return vq_argmin(flat_input.detach(), self.embedding.weight.detach(), self._e_sqnorm())

    def get_code_indices_channel_major(self, z, hist=None):
        """Indices for an encoder output z[B, D, S] without the permute/contiguous copy."""
        return vq_argmin(z.detach(), self.embedding.weight.detach(), self._e_sqnorm(), channel_major=True, hist=hist)

    @_lib.on_tensor_device
    def forward(self, inputs):
        input_shape = inputs.shape
        flat_input = inputs.reshape(-1, self.embedding_dim)
        w = self.embedding.weight
        K = self.num_embeddings
        hist = torch.zeros((K,), dtype=torch.int64, device=flat_input.device)
        idx = vq_argmin(flat_input.detach(), w.detach(), self._e_sqnorm(), hist=hist)

        grad_path = torch.is_grad_enabled() and (inputs.requires_grad or w.requires_grad)
        if self.training or grad_path:
            # differentiable gather with the PRE-update codebook (training/model.py:100)
            quantized = F.embedding(idx, w).view(input_shape)
        if self.training:
            # EMA codebook update (training/model.py:103-126): deterministic segmented sums instead of the
            # one-hot matmuls, then the in-place EMA kernel.  With `ema_group` set (one process per GPU), the
            # (counts, dw) pair is all-reduced first, so every rank applies the same global update.
            with torch.no_grad():
                dw, counts = vq_ema_sums(flat_input.detach(), idx, K)
                group = getattr(self, "ema_group", None)
                if group is not None:
                    import torch.distributed as dist
                    dist.all_reduce(counts, group=group)
                    dist.all_reduce(dw, group=group)
                vq_ema_apply(self.ema_cluster_size, self.ema_w, self.embedding.weight.data, dw, counts, self.decay,
                             self.epsilon)
                # the kernel rewrote the codebook through its raw pointer (no autograd version bump): the cached |e|^2
                # belongs to the old rows (the reference recomputes sum(weight**2) every call, training/model.py:91)
                self._ee_key = self._ee = None
                self._cb_epoch += 1
        if self.training or grad_path:
            e_latent_loss = F.mse_loss(quantized.detach(), inputs)
            q_latent_loss = F.mse_loss(quantized, inputs.detach())
            loss = q_latent_loss + self.commitment_cost * e_latent_loss
            quantized = inputs + (quantized - inputs).detach()
        else:
            q, err = vq_gather_loss(flat_input.detach(), w.detach(), idx)
            mse = (err[0] / flat_input.numel()).to(torch.float32)
            loss = mse + self.commitment_cost * mse
            quantized = q.view(input_shape)
        perplexity = perplexity_from_hist(hist)
        return loss, quantized, perplexity, idx.view(input_shape[0], -1)


class VQVAE(nn.Module):
    def __init__(self, input_channels, embedding_dim, num_embeddings, commitment_cost, decay):
        super().__init__()
        self.input_channels = input_channels
        self.embedding_dim = embedding_dim
        self.encoder = nn.Sequential(
            nn.Conv1d(input_channels, 64, kernel_size=4, stride=2, padding=1), nn.ReLU(),
            nn.Conv1d(64, 128, kernel_size=4, stride=2, padding=1), nn.ReLU(),
            nn.Conv1d(128, 256, kernel_size=3, stride=1, padding=1), nn.ReLU(),
            nn.Conv1d(256, embedding_dim, kernel_size=1, stride=1))
        self.vq_layer = VectorQuantizerEMA(num_embeddings, embedding_dim, commitment_cost, decay)
        self.decoder = nn.Sequential(
            nn.ConvTranspose1d(embedding_dim, 256, kernel_size=1, stride=1), nn.ReLU(),
            nn.ConvTranspose1d(256, 128, kernel_size=3, stride=1, padding=1), nn.ReLU(),
            nn.ConvTranspose1d(128, 64, kernel_size=4, stride=2, padding=1, output_padding=1), nn.ReLU(),
            nn.ConvTranspose1d(64, input_channels, kernel_size=4, stride=2, padding=1, output_padding=1))

    def forward(self, x):
        """Training/eval forward with reconstruction (training/model.py:195-203).  The encoder/decoder run
        through torch autograd here (training is outside the hot path); the quantizer uses the HIP kernels."""
        z = self.encoder(x)
        z_permuted = z.permute(0, 2, 1).contiguous()
        vq_loss, quantized, perplexity, indices = self.vq_layer(z_permuted)
        x_recon = self.decoder(quantized.permute(0, 2, 1).contiguous())
        return vq_loss, x_recon, perplexity, indices

    def _folded_codebook(self):
        """(codebook', e_sqnorm') of ``fold_projection`` for the current weights, recomputed when the codebook or the last encoder
        layer changes (keyed on storage pointer + autograd version + the quantizer's in-place EMA update counter)."""
        w, pw, pb = self.vq_layer.embedding.weight, self.encoder[6].weight, self.encoder[6].bias
        key = (w.data_ptr(), w._version, pw.data_ptr(), pw._version, pb.data_ptr(), pb._version, str(w.device), self.vq_layer._cb_epoch)
        if getattr(self, "_fold_key", None) != key:
            self._fold = fold_projection(w.detach(), pw.detach(), pb.detach())
            self._fold_key = key
        return self._fold

    def _screen_set(self, folded: bool):
        """What the screened search runs on, rebuilt when the codebook changes: the (folded) codebook WITHOUT rows that repeat an
        earlier row bit for bit (``dedup_codebook``: a first-minimum argmin never returns them; the shipped checkpoint has 100
        all-zero rows), its constants, its bf16 piece table, and the map back to code numbers (None: nothing was dropped)."""
        w = self.vq_layer.embedding.weight
        if folded:
            self._folded_codebook()      # refreshes _fold_key
        key = (folded, w.data_ptr(), w._version, str(w.device), self.vq_layer._cb_epoch,
               getattr(self, "_fold_key", None) if folded else None)
        if getattr(self, "_screen_key", None) != key:
            cb, ee = self._folded_codebook() if folded else (w.detach(), self.vq_layer._e_sqnorm())
            r = dedup_codebook(cb, ee)
            cmap, kc = None, int(cb.shape[0])
            if r is not None:
                cb, ee, cmap, kc = r
            self._screen = (cb, ee, vq_split_tables(cb), cmap, kc)
            self._screen_key = key
        return self._screen

    def _screened_indices(self, h, folded: bool, hist):
        cb, ee, tab, cmap, kc = self._screen_set(folded)
        if cmap is None:
            return vq_argmin(h, cb, ee, channel_major=True, no_xnorm=folded, hist=hist, split_tables=tab, screened=True)
        kp = int(cb.shape[0])    # remap over the padding rows too (they map to code 0: what the full search returns where every score is +inf)
        hist_s = None            # the remap pass counts the tokens itself (kp <= 12288), else the search counts over searched rows
        if hist is not None and kp > 12288:
            hist_s = torch.zeros((kp,), dtype=torch.int64, device=hist.device)
        idx = vq_argmin(h, cb, ee, channel_major=True, no_xnorm=folded, hist=hist_s, split_tables=tab, screened=True)
        return remap_codes(idx, cmap, kp, hist_s, hist)

    @torch.no_grad()
    @_lib.on_tensor_device
    def encode(self, x, hist=None, fold_projection: bool = False, screened: bool = False):
        """Input x[B, C, L] -> code indices int64 [B, L//4] (training/model.py:206-214), all on HIP kernels;
        the encoder output is consumed channel-major, so the permute(0,2,1).contiguous() copy disappears.
        ``fold_projection=True``: the last 1x1 convolution is folded into the codebook (see ``fold_projection``): its GEMM and
        the z tensor disappear; indices equal the default's except at documented fp32 near-ties.
        ``screened=True``: the same indices as without it, bit for bit, from the screened search (``lbm_vq_argmin_screened_f32``:
        2.4x faster at the shipped sizes)."""
        if fold_projection:
            h = x
            for i, (st, pd) in zip((0, 2, 4), ((2, 1), (2, 1), (1, 1))):
                h = conv1d(h, self.encoder[i].weight, self.encoder[i].bias, st, pd, True, pitched_out=True)
            if screened:
                idx = self._screened_indices(h, True, hist)
            else:
                cbf, eef = self._folded_codebook()
                idx = vq_argmin(h, cbf, eef, channel_major=True, no_xnorm=True, hist=hist)
            return idx.view(x.shape[0], -1)
        z = encoder_forward(x, self.encoder, pitched_out=True)
        if screened:
            idx = self._screened_indices(z.detach(), False, hist)
        else:
            idx = self.vq_layer.get_code_indices_channel_major(z, hist=hist)
        return idx.view(x.shape[0], -1)


def encode_data_for_transformer(vq_model, dataloader, device, fold_projection: bool = False, screened: bool = False):
    """training/model.py:411-427: returns (np.int64 [E, S] indices, np labels).  Batches are encoded
    back-to-back on the device; indices come back to the host once, after the loop.  ``fold_projection`` and ``screened`` (not
    reference arguments) are handed to ``VQVAE.encode``."""
    vq_model.eval()
    all_indices, all_labels = [], []
    with torch.no_grad():
        for data, labels in dataloader:
            data = data.to(device, non_blocking=True)
            all_indices.append(vq_model.encode(data, fold_projection=fold_projection, screened=screened))
            all_labels.append(labels.numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels))
    all_indices = torch.cat(all_indices, dim=0).cpu().numpy()
    all_labels = np.concatenate(all_labels, axis=0)
    return all_indices, all_labels
