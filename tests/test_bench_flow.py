"""bench.py's own flows on a box without a GPU: the kernels' host emulation (tests/hostemu_binding) handed in as the library.
What is checked is the FILE -- argument handling, the single-process N-GPU mode the driver starts (`python bench.py --gpus N`:
N handles stepped by swimsim_cluster_step), the JSON contract, the oracle check inside the run -- not a number."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_bench(argv, monkeypatch, **env):
    import bench
    from tests import hostemu_binding
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(bench, "PREROLL_MAX", 60)
    monkeypatch.setattr(bench, "SATURATED_D", 0.0)        # (a 2 048-member cluster never carries the million-member load: take what it has)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main(argv, abi=hostemu_binding.load())
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, buf.getvalue()
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_bench_line_single_process(monkeypatch, gpus):
    out = run_bench(["--gpus", str(gpus), "--steps", "6", "--warmup", "3", "--members", "512"], monkeypatch, SWIM_BENCH_SHARE_GPU="1")
    assert out["n_gpus"] == gpus and out["steps"] == 6 and out["warmup"] == 3
    assert out["unit"] == "member-ticks/s" and out["value"] > 0 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["config"]["members_per_gpu"] == 512 and "workload" in out["config"]
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0
    assert out["verified_vs_oracle"] is True and out["cpu_baseline"]["kind"] == "port"
    if gpus > 1:
        ex = out["exchange"]
        assert ex["bytes_per_gpu_per_tick"] >= 512 * 9 * (gpus - 1) and ex["xgmi_peak_GBs"] == 7 * 153.0
        assert "ONE process" in out["config"]["parallelism"]



def test_default_run_carries_baseline_row_3s_as_written(monkeypatch):
    """The default one-GPU run measures TWO clusters: the headline (1 crash per tick) and BASELINE.md row 3(s) as
    written (9.5 per tick, settling, 100 warm-up ticks) -- each with its own roofline and oracle check."""
    import bench
    monkeypatch.setattr(bench, "N_MEMBERS", 2048)
    out = run_bench(["--steps", "6", "--warmup", "3", "--members", "2048"], monkeypatch)
    assert out["verified_vs_oracle"] is True and "settling" not in out["config"]["workload"]
    w = out["config3s_as_written"]
    assert w["verified_vs_oracle"] is True and w["cpu_baseline"]["kind"] == "port"
    assert w["steps"] == 6 and w["warmup"] == 100 and "9.5 crashes per tick" in w["config"]["workload"] and "settling" in w["config"]["workload"]
    assert w["roofline"]["bound"] == "hbm" and set(w["roofline"]["kernels"]) == {"probe_kernel", "merge_kernel"}
    assert set(w["per_member_tick"]) == {"d", "r", "c", "f"} and w["value"] > 0 and w["ms_per_step"] > 0
    out2 = run_bench(["--steps", "6", "--warmup", "3", "--members", "2048", "--no-as-written"], monkeypatch)
    assert "config3s_as_written" not in out2
