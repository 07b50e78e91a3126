"""Data-set drivers (multi-train/withpsd.py:70-91,326-414; preprocessing_unsupervised.py:30-123) on CPU: file loop with per-file
skip, file-order concatenation, output files, data-set median -- serial and sharded by recording over a world-size-2 gloo group.
The per-file bodies here are the ORACLE's (the product's bodies need the GPU: tests/test_gpu_dataset.py)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import lbm_amd
from tests import dataset_util as du


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_classify_channels_follows_the_script():
    names = ["Fz", "veog", "HEOG", "EKG", "A1", "tp10", "Vehicle Position", "serial in", "Cz", "IO", "Reference", "O1"]
    got = lbm_amd.classify_channels(names)
    assert got == {"eeg": [0, 8, 11], "eog": [1, 2, 9], "ecg": [3], "misc": [4, 5, 6, 7, 10]}


def test_withpsd_dataset_serial(tmp_path):
    files = du.make_withpsd_dataset(str(tmp_path / "data"))
    out = lbm_amd.preprocess_dataset_withpsd(data_dir=str(tmp_path / "data"), output_dir=str(tmp_path / "out"),
                                             process_file=du.oracle_withpsd_file)
    status = dict((os.path.basename(p), s) for p, s in out["files"])
    assert set(status) == {os.path.basename(f) for f in files}
    assert isinstance(status["rec_bad.set"], str) and isinstance(status["rec_noann.set"], str) and "SkipFile" in status["rec_noann.set"]
    good = [p for p, s in out["files"] if not isinstance(s, str)]
    assert len(good) == 5
    # file-order concatenation of exactly what the per-file body returns (np.concatenate of the reference's lists)
    per = [du.oracle_withpsd_file(p) for p in good]
    np.testing.assert_array_equal(out["time_epochs"], np.concatenate([r["time_epochs"] for r in per]))
    np.testing.assert_array_equal(out["psd_epochs"], np.concatenate([r["psd_epochs"] for r in per]))
    assert out["labels"].tolist() == np.concatenate([r["labels"] for r in per]).tolist()
    assert [s for _, s in out["files"] if not isinstance(s, str)] == [len(r["labels"]) for r in per]
    # the reference's four files, float64
    base = os.path.join(str(tmp_path / "out"), "combined_s02_stimlocked_std_rej_psd")
    assert out["paths"] == [base + s for s in ("_time_epochs.npy", "_psd_epochs.npy", "_psd_freqs.npy", "_labels.npy")]
    t, p, f, l_ = (np.load(q) for q in out["paths"])
    assert t.dtype == np.float64 and p.dtype == np.float64 and t.shape[0] == p.shape[0] == l_.shape[0] and p.shape[2] == f.shape[0] == 27
    np.testing.assert_array_equal(t, out["time_epochs"])
    assert set(l_.tolist()) <= {"alert", "transition", "drowsy"}
    # nothing usable: nothing saved, arrays None (the script prints an error and skips saving)
    none = lbm_amd.preprocess_dataset_withpsd(files=[str(tmp_path / "data" / "rec_bad.set")], output_dir=str(tmp_path / "out2"),
                                              process_file=du.oracle_withpsd_file)
    assert none["time_epochs"] is None and none["paths"] == [] and not os.path.exists(tmp_path / "out2")


def _worker(rank, world, port, data_dir, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 1 is handed a DIFFERENT (reversed) list: rank 0's order must win, or the shards would overlap
        files = sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir) if f.endswith(".set"))
        w = lbm_amd.preprocess_dataset_withpsd(files=files if rank == 0 else files[::-1], output_dir=os.path.join(out_dir, "w"),
                                               process_file=du.oracle_withpsd_file)
        u = lbm_amd.process_unsupervised_dataset(files=[f for f in (files if rank == 0 else files[::-1])],
                                                 output_path=os.path.join(out_dir, "u"), process_file=du.oracle_unsupervised_file)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), time=w["time_epochs"], psd=w["psd_epochs"], labels=w["labels"],
                 X=u["X"], y=u["y"], thr=u["drowsy_threshold"], n_files=len(w["files"]))
    finally:
        dist.destroy_process_group()


def test_datasets_sharded_by_recording_world2(tmp_path):
    data_dir = str(tmp_path / "data")
    files = du.make_withpsd_dataset(data_dir)
    serial_w = lbm_amd.preprocess_dataset_withpsd(files=files, process_file=du.oracle_withpsd_file)
    serial_u = lbm_amd.process_unsupervised_dataset(files=files, process_file=du.oracle_unsupervised_file)
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    mp.spawn(_worker, args=(2, _free_port(), data_dir, out_dir), nprocs=2, join=True)
    for rank in range(2):
        with np.load(os.path.join(out_dir, f"rank{rank}.npz")) as r:       # every rank holds the whole data set, in file order
            np.testing.assert_array_equal(r["time"], serial_w["time_epochs"])
            np.testing.assert_array_equal(r["psd"], serial_w["psd_epochs"])
            assert r["labels"].tolist() == serial_w["labels"].tolist() and int(r["n_files"]) == len(files)
            np.testing.assert_array_equal(r["X"], serial_u["X"])
            np.testing.assert_array_equal(r["y"], serial_u["y"])
            assert float(r["thr"]) == serial_u["drowsy_threshold"]
    # rank 0 wrote the reference's files
    base = os.path.join(out_dir, "w", "combined_s02_stimlocked_std_rej_psd")
    np.testing.assert_array_equal(np.load(base + "_time_epochs.npy"), serial_w["time_epochs"])
    np.testing.assert_array_equal(np.load(os.path.join(out_dir, "u", "X_psd_multi.npy")), serial_u["X"])
    np.testing.assert_array_equal(np.load(os.path.join(out_dir, "u", "y_label_multi.npy")), serial_u["y"])


def test_unsupervised_dataset_serial_median_and_labels(tmp_path):
    files = du.make_withpsd_dataset(str(tmp_path / "data"), n_files=4)
    out = lbm_amd.process_unsupervised_dataset(dataset_path=str(tmp_path / "data"), output_path=str(tmp_path / "out"),
                                               process_file=du.oracle_unsupervised_file)
    status = dict((os.path.basename(p), s) for p, s in out["files"])
    assert status["rec_bad.set"] != "ok" and sum(s == "ok" for s in status.values()) == 5      # the file without events is fine here
    assert out["X"].shape == (5, 4, 53) and out["y"].shape == (5,)
    thr = float(np.median(out["theta"] / out["beta"]))
    assert out["drowsy_threshold"] == thr
    assert out["y"].tolist() == [lbm_amd.classify_eeg_state(t, a, b, thr) for t, a, b in zip(out["theta"], out["alpha"], out["beta"])]
    np.testing.assert_array_equal(np.load(out["paths"][0]), out["X"])
    np.testing.assert_array_equal(np.load(out["paths"][1]), out["y"])
    # os.listdir order is what the script walks: the report follows it
    want = [f for f in os.listdir(str(tmp_path / "data")) if f.endswith(".set")]
    assert [os.path.basename(p) for p, _ in out["files"]] == want
