"""bench.py --gpus N: the launcher's argument / environment plumbing (CPU; the rank processes here are a stand-in script)."""
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


STUB = textwrap.dedent("""
    import json, os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if "--fail-rank" in sys.argv and r == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(3)
    if "--hang-others" in sys.argv and r != int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        time.sleep(60)            # a rank waiting in a collective for the one that died
    if r == 0:
        print("some library chatter")
        print(json.dumps({"n_gpus": w, "argv": sys.argv[1:]}))
    """)


@pytest.fixture
def stub(tmp_path):
    p = tmp_path / "rank_stub.py"
    p.write_text(STUB)
    return str(p)


def test_rank_env_is_what_torchrun_sets():
    env = bench.rank_env({"PATH": "/bin", "WORLD_SIZE": "9"}, 3, 8, 29512)
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29512" and env["PATH"] == "/bin"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.rank_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # caller's choice kept


def test_launcher_starts_n_ranks_and_relays_rank0_json(stub, capfd):
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "2"], script=stub)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1                                       # ONE JSON line on stdout; chatter goes to stderr
    assert json.loads(lines[0]) == {"n_gpus": 4, "argv": ["--gpus", "4", "--steps", "2"]}
    assert "some library chatter" in err


def test_launcher_fails_when_any_rank_fails_and_stops_the_others(stub, capfd):
    import time
    t0 = time.time()
    rc = bench.launch_ranks(3, ["--fail-rank", "1", "--hang-others"], script=stub)
    assert rc != 0 and time.time() - t0 < 30                     # did not wait for the hung ranks
    assert "rank 1 exited with code 3" in capfd.readouterr().err


def test_launcher_fails_without_a_json_line(stub, capfd):
    assert bench.launch_ranks(2, ["--fail-rank", "0"], script=stub) != 0


def test_gpus_flag_must_match_world_size(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "1"])
    assert "WORLD_SIZE=2" in str(e.value)


def test_gpus_flag_launches_when_no_world_size(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda world, argv: seen.update(world=world, argv=argv) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8", "--steps", "1"])
    assert e.value.code == 0 and seen == {"world": 8, "argv": ["--gpus", "8", "--steps", "1"]}


def test_parent_module_import_does_not_load_the_hip_library():
    """The launcher parent must stay free of GPU state: importing bench must not import the product package."""
    import subprocess
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'lbm_amd' not in sys.modules and bench.lbm_amd is None" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_stage_report_executed_flops_and_peaks():
    """`stages` of the bench line: algorithmic rate against the fp32 matrix peak; where a kernel executes something else (Winograd,
    the bf16 screen) the executed rate against the peak of the unit it ran on, and for the bf16 unit that is what `frac` reports."""
    import bench
    m = bench.stage_model(16384, 30, 1500, 825, 512, 256, 375, 750, 27, vq_method="screen")
    assert m["vq"]["exec_flops"] == 3 * m["vq"]["flops"] and m["vq"]["peak"] == bench.PEAK_BF16_TFLOPS
    assert abs(m["conv3"]["exec_flops"] / m["conv3"]["flops"] - 2.0 / 3.0) < 1e-12 and "peak" not in m["conv3"]
    assert "exec_flops" not in bench.stage_model(16384, 30, 1500, 825, 512, 256, 375, 750, 27)["vq"]
    rep = bench.stage_report({"vq": [5.0], "conv3": [8.0], "welch": [0.8]}, m)
    vq, c3, w = rep["vq"], rep["conv3"], rep["welch"]
    assert abs(vq["executed_tflops"] - 3 * vq["tflops"]) < 0.01 and vq["frac"] == vq["executed_frac"] == round(vq["executed_tflops"] / 2500.0, 4)
    assert abs(c3["frac"] - c3["tflops"] / bench.PEAK_FP32_TFLOPS) < 1e-4 and abs(c3["executed_frac"] - c3["frac"] * 2 / 3) < 1e-3
    assert w["bound"] == "hbm" and abs(w["frac"] - w["gbs"] / bench.PEAK_HBM_GBS) < 1e-4 and "executed_frac" not in w


def test_chunk_roofline_is_floor_over_measured():
    """`roofline.chunk`: one end-to-end fraction next to the dominant kernel's (HBM stages at their bytes, matrix stages at their
    EXECUTED flops against the unit they ran on)."""
    import bench
    m = bench.stage_model(16384, 30, 1500, 825, 512, 256, 375, 750, 27, vq_method="screen")
    ms = {"fir": 3.0, "affine_post": 1.5, "welch": 0.85, "conv3": 8.0, "vq": 4.7}
    rep = bench.stage_report({k: [v] for k, v in ms.items()}, m)
    c = bench.chunk_roofline(rep, m)
    want = (m["fir"]["bytes"] + m["affine_post"]["bytes"] + m["welch"]["bytes"]) / 8e12 * 1e3 \
        + m["conv3"]["exec_flops"] / 157.3e12 * 1e3 + m["vq"]["exec_flops"] / 2500e12 * 1e3
    assert abs(c["floor_ms"] - want) < 1e-3 and abs(c["measured_ms"] - sum(ms.values())) < 1e-9
    assert abs(c["frac"] - want / sum(ms.values())) < 1e-3 and c["floor_ms_attainable"] > c["floor_ms"] and 0 < c["frac"] < 1
