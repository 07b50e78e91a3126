"""Data-set level drivers: the loops that turn a directory of recordings into the reference's output files.

``preprocess_dataset_withpsd``  = multi-train/withpsd.py:70-91 (glob, per-file ``try ... continue``) + :98-110 (channel typing)
+ the per-file body (``recording.preprocess_recording_withpsd``) + :279-305 (collect aligned time / PSD / label arrays)
+ :326-414 (concatenate in file order, four ``np.save``s).
``process_unsupervised_dataset`` = preprocessing_unsupervised.py:30-67 (pass 1: band powers per file, data-set median of
theta/beta) + :86-123 (pass 2: labels, ``X_psd_multi.npy`` / ``y_label_multi.npy``).

Multi-GPU: recordings are independent, so the FILES shard across ranks (contiguous blocks of the file list, one process per GPU);
the only exchanges are an all-gather of the per-file results in rank order (= file order, what the reference's lists hold) and,
for the unsupervised flow, of one theta/beta ratio per file for the median.  Every rank returns the whole data set; rank 0 writes
the files.  The per-file body is injectable (``process_file``) so that the orchestration is testable without a GPU.
"""
from __future__ import annotations

import glob
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .constants import FILTER_HFREQ, FILTER_LFREQ
from .pipeline import shard_range
from .recording import VIGILANCE_EVENT_IDS, read_raw_eeglab

EOG_CHANNEL_NAMES = ["VEOG", "HEOG", "EOG", "IO"]        # multi-train/withpsd.py:21
ECG_CHANNEL_NAMES = ["ECG", "EKG"]                       # :22
MISC_CHANNEL_PREFIXES = ["position", "vehicle", "serial", "event"]                                     # :106
REFERENCE_CHANNEL_NAMES = ["M1", "M2", "A1", "A2", "TP9", "TP10", "REF", "Reference", "Mastoid"]       # :107
OUTPUT_FILENAME_BASE = "combined_s02_stimlocked_std_rej_psd"                                           # :18


class SkipFile(Exception):
    """A recording the reference's loop skips with a warning and ``continue`` (no EEG channels, no annotations, no trials, no
    epochs left after rejection ...)."""


def classify_channels(ch_names: Sequence[str], eog_names=EOG_CHANNEL_NAMES, ecg_names=ECG_CHANNEL_NAMES) -> Dict[str, List[int]]:
    """Channel typing of multi-train/withpsd.py:101-110: EOG / ECG by (case-insensitive) name, 'misc' for vehicle / position /
    serial / event traces and reference electrodes, everything else EEG."""
    eog_u, ecg_u = [n.upper() for n in eog_names], [n.upper() for n in ecg_names]
    out = {"eeg": [], "eog": [], "ecg": [], "misc": []}
    for i, ch in enumerate(ch_names):
        if ch.upper() in eog_u:
            out["eog"].append(i)
        elif ch.upper() in ecg_u:
            out["ecg"].append(i)
        elif any(p in ch.lower() for p in MISC_CHANNEL_PREFIXES) or any(r.lower() == ch.lower() for r in REFERENCE_CHANNEL_NAMES):
            out["misc"].append(i)
        else:
            out["eeg"].append(i)
    return out


# ---------------------------------------------------------------------------------------------- exchange of per-file results
def _world(group):
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(group), dist.get_world_size(group)


def _agree_on_files(files: List[str], group) -> List[str]:
    """Every rank must walk the same list: rank 0's (``glob`` / ``os.listdir`` order is the file system's) is broadcast."""
    dist, rank, world = _world(group)
    if world == 1:
        return files
    box = [files if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return list(box[0])


def _gather_in_file_order(mine: list, group) -> list:
    """Per-file results of every rank, concatenated in rank order = file order (contiguous file blocks per rank).  The results
    are small host objects per recording (arrays of one file's epochs at most), gathered as objects; the sample passes that
    produced them ran on each rank's GPU."""
    dist, rank, world = _world(group)
    if world == 1:
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return [r for part in parts for r in part]


# ---------------------------------------------------------------------------------------------- multi-train/withpsd.py
def withpsd_file(path: str, reader: Callable = read_raw_eeglab, **recording_kw) -> dict:
    """One pass of the reference's loop body (multi-train/withpsd.py:92-305) on one ``.set`` file, every sample pass on the GPU.
    Returns host arrays ``dict(time_epochs [E, C, Tw] f64, psd_epochs [E, C, F] f64, psd_freqs [F], labels [E] str)``; raises
    ``SkipFile`` where the script prints a warning and continues."""
    from .recording import preprocess_recording_withpsd
    raw = reader(path)                                                                # :96
    picks = classify_channels(raw["ch_names"])                                        # :101-111
    if not picks["eeg"]:
        raise SkipFile("no EEG channels found")                                       # :126
    if not raw["annotations"]:
        raise SkipFile("no annotations found")                                        # :158
    data = np.asarray(raw["data"])
    eog = data[picks["eog"]] if picks["eog"] else None                                # :135-139
    ecg = data[picks["ecg"][0]] if picks["ecg"] else None                             # :142-145 (one channel, else MNE's first)
    out = preprocess_recording_withpsd(data[picks["eeg"]], raw["sfreq"], raw["annotations"], eog=eog, ecg=ecg, **recording_kw)
    if len(out["labels"]) == 0:
        raise SkipFile("no epochs remaining after rejection")                         # :164-225 (each of them a `continue`)
    host = lambda t: np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64)  # noqa: E731
    return dict(time_epochs=host(out["time_epochs"]), psd_epochs=host(out["psd_epochs"]), psd_freqs=np.asarray(out["psd_freqs"]),
                labels=np.asarray(out["labels"]))


def preprocess_dataset_withpsd(files: Optional[Sequence[str]] = None, *, data_dir: Optional[str] = None, output_dir: Optional[str] = None,
                               output_filename_base: str = OUTPUT_FILENAME_BASE, group=None, process_file: Callable = withpsd_file,
                               verbose: bool = False, **recording_kw) -> dict:
    """multi-train/withpsd.py as a function: every ``*.set`` of ``data_dir`` (or the given ``files``) through the per-file body, a
    failing file skipped (:314-320), the kept epochs of all files concatenated in file order (:336, :360, :381) and -- with
    ``output_dir`` -- saved as ``<base>_time_epochs.npy``, ``<base>_psd_epochs.npy``, ``<base>_psd_freqs.npy``, ``<base>_labels.npy``
    (:61-66, :395-414; float64 arrays like the reference's).

    With a process group (``group`` or the default one) the files are sharded over the ranks in contiguous blocks; every rank
    returns the whole result, rank 0 saves.  Returns ``dict(time_epochs, psd_epochs, psd_freqs, labels, files=[(path, n_epochs
    or error text)], paths)``; the arrays are ``None`` when no file produced epochs (the script then saves nothing).
    """
    if files is None:
        if data_dir is None:
            raise ValueError("preprocess_dataset_withpsd: give files or data_dir")
        files = glob.glob(os.path.join(data_dir, "*.set"))                            # :70
    files = _agree_on_files(list(files), group)
    if not files:
        raise FileNotFoundError(f"No .set files found in directory: {data_dir}")      # :71-73 (the script exits)
    dist, rank, world = _world(group)
    lo, hi = shard_range(len(files), rank, world)
    mine = []
    for path in files[lo:hi]:                                                         # :87
        try:
            r = process_file(path, **recording_kw)
            n = len(r["labels"])
            if not (len(r["time_epochs"]) == n and len(r["psd_epochs"]) == n):        # :286, :305
                raise SkipFile(f"mismatch epochs/labels/PSD: {len(r['time_epochs'])}, {n}, {len(r['psd_epochs'])}")
            ids = np.asarray([VIGILANCE_EVENT_IDS[str(l_)] for l_ in r["labels"]], dtype=np.int64)
            mine.append(dict(path=path, ok=True, time=np.asarray(r["time_epochs"], dtype=np.float64),
                             psd=np.asarray(r["psd_epochs"], dtype=np.float64), freqs=np.asarray(r["psd_freqs"]), ids=ids))
        except Exception as e:                                                        # noqa: BLE001  (:314-320: skip this file)
            if verbose:
                print(f"   ERROR processing file {os.path.basename(path)}: {e}\n   Skipping this file.")
            mine.append(dict(path=path, ok=False, error=f"{type(e).__name__}: {e}"))
    allr = _gather_in_file_order(mine, group)
    good = [r for r in allr if r["ok"]]
    report = [(r["path"], len(r["ids"]) if r["ok"] else r["error"]) for r in allr]
    res = dict(time_epochs=None, psd_epochs=None, psd_freqs=None, labels=None, files=report, paths=[])
    if not good:
        return res
    rev = {v: k for k, v in VIGILANCE_EVENT_IDS.items()}                              # :281
    res["psd_freqs"] = good[0]["freqs"]                                               # :253-254: the first file's bins
    res["time_epochs"] = np.concatenate([r["time"] for r in good], axis=0)            # :336 (raises on inconsistent shapes, as there)
    res["psd_epochs"] = np.concatenate([r["psd"] for r in good], axis=0)              # :360
    res["labels"] = np.asarray([rev[int(i)] for r in good for i in r["ids"]])         # :283, :381
    if output_dir is not None and rank == 0:
        from .epochs import save_withpsd_outputs
        os.makedirs(output_dir, exist_ok=True)                                        # :59
        res["paths"] = save_withpsd_outputs(os.path.join(output_dir, output_filename_base), res["time_epochs"], res["psd_epochs"],
                                            res["psd_freqs"], res["labels"])
    if world > 1:
        dist.barrier(group=group)                                                     # the files exist when any rank returns
    return res


# ---------------------------------------------------------------------------------------------- preprocessing_unsupervised.py
def unsupervised_file(path: str, reader: Callable = read_raw_eeglab, **recording_kw) -> dict:
    """preprocessing_unsupervised.py:34-61 on one ``.set`` file (every channel of the file, as ``ica.fit(raw)`` /
    ``raw.compute_psd`` take them).  Returns ``dict(theta, alpha, beta, log_psds [C, F] f64)``."""
    from .recording import process_unsupervised_recording
    raw = reader(path)                                                                # :34
    out = process_unsupervised_recording(raw["data"], raw["sfreq"], **recording_kw)   # :35-61
    log_psds = 10.0 * np.log10(out["psds"].detach().cpu().numpy().astype(np.float64))   # :45
    return dict(theta=float(out["theta"]), alpha=float(out["alpha"]), beta=float(out["beta"]), log_psds=log_psds)


def process_unsupervised_dataset(files: Optional[Sequence[str]] = None, *, dataset_path: Optional[str] = None,
                                 output_path: Optional[str] = None, group=None, process_file: Callable = unsupervised_file,
                                 verbose: bool = False, **recording_kw) -> dict:
    """preprocessing_unsupervised.py as a function.  Pass 1 (:30-64): every ``.set`` file of ``dataset_path`` (``os.listdir``
    order) -> theta / alpha / beta dB means; a failing file is reported and left out.  Threshold = median of theta/beta over the
    data set (:66-67).  Pass 2 (:86-120) recomputes the same per-file quantities in the reference (deterministic:
    ``random_state=42``); here pass 1's are kept, and each file gets ``classify_eeg_state(theta, alpha, beta, threshold)`` (:70-84).
    Saves ``X_psd_multi.npy`` (log-PSD [n_files, C, F]) and ``y_label_multi.npy`` under ``output_path`` (:122-123).

    With a process group the files shard over the ranks; the per-file band powers (one theta/beta ratio per file for the median)
    and log-PSDs are all-gathered in file order.  Returns ``dict(X, y, drowsy_threshold, theta, alpha, beta, files, paths)``.
    """
    from .psd import classify_eeg_state
    if files is None:
        if dataset_path is None:
            raise ValueError("process_unsupervised_dataset: give files or dataset_path")
        files = [os.path.join(dataset_path, f) for f in os.listdir(dataset_path) if f.endswith(".set")]   # :30-32
    files = _agree_on_files(list(files), group)
    dist, rank, world = _world(group)
    lo, hi = shard_range(len(files), rank, world)
    mine = []
    for path in files[lo:hi]:
        try:
            r = process_file(path, **recording_kw)
            mine.append(dict(path=path, ok=True, theta=float(r["theta"]), alpha=float(r["alpha"]), beta=float(r["beta"]),
                             log_psds=np.asarray(r["log_psds"], dtype=np.float64)))
        except Exception as e:                                                        # noqa: BLE001  (:63-64, :119-120)
            if verbose:
                print(f" Error reading {os.path.basename(path)}: {e}")
            mine.append(dict(path=path, ok=False, error=f"{type(e).__name__}: {e}"))
    allr = _gather_in_file_order(mine, group)
    good = [r for r in allr if r["ok"]]
    report = [(r["path"], "ok" if r["ok"] else r["error"]) for r in allr]
    theta, alpha, beta = (np.asarray([r[k] for r in good], dtype=np.float64) for k in ("theta", "alpha", "beta"))
    res = dict(X=None, y=None, drowsy_threshold=float("nan"), theta=theta, alpha=alpha, beta=beta, files=report, paths=[])
    if not good:
        return res
    thr = float(np.median(theta / beta))                                              # :61, :66-67
    res["drowsy_threshold"] = thr
    res["y"] = np.array([classify_eeg_state(t, a, b, thr) for t, a, b in zip(theta, alpha, beta)])      # :112-115
    res["X"] = np.array([r["log_psds"] for r in good])                                # :114, :122
    if output_path is not None and rank == 0:
        os.makedirs(output_path, exist_ok=True)                                       # :16
        res["paths"] = [os.path.join(output_path, "X_psd_multi.npy"), os.path.join(output_path, "y_label_multi.npy")]
        np.save(res["paths"][0], res["X"])                                            # :122
        np.save(res["paths"][1], res["y"])                                            # :123
    if world > 1:
        dist.barrier(group=group)
    return res
