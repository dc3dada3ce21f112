"""bench.py keeps its contract with the driver: one JSON line on stdout with the agreed fields, for every
workload (small sizes here; the numbers are not checked, the shape of the line is)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config", "roofline")


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(d, steps=3, warmup=1):
    for k in COMMON:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_chamfer_line(cuda):
    d = _run("--batch", "8", "--points", "4096")
    _check_common(d)
    assert d["metric"] == "chamfer_fwd_bwd_point_pairs_per_s" and d["unit"] == "pairs/s"
    modes = d["launch_modes_ms_per_step"]
    assert {"ext", "eager"} <= set(modes) and all(modes[k] > 0 for k in ("ext", "eager"))
    # the timed value is whole-job pairs per second of the step it reports
    assert abs(d["value"] - 2.0 * 8 * 4096 * 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    base = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in base, k
    assert base["kind"] == "port" and base["cores"] >= 1


def test_chamfer_launch_modes_agree_on_the_shape_of_the_line(cuda):
    for mode in ("graph", "ext", "eager"):
        d = _run("--batch", "8", "--points", "4096", "--launch", mode, "--no-cpu-baseline")
        _check_common(d)
        assert "cpu_baseline" not in d


def test_fps_and_ball_group_lines(cuda):
    d = _run("--workload", "fps", "--batch", "4", "--points", "8192")
    _check_common(d)
    assert d["metric"] == "fps_point_updates_per_s"
    d = _run("--workload", "ball_group")
    _check_common(d)
    assert d["metric"] == "group_points_output_bytes_per_s" and 0.0 < d["roofline"]["frac"] < 1.0


def test_default_line_is_the_autograd_operator_and_carries_configs_3_and_4(cuda):
    """`value` is timed on the torch.autograd.Function path; the default command also runs
    BASELINE.json configs 3 and 4 briefly, each with its own roofline, and the other point distributions."""
    d = _run("--no-cpu-baseline")
    _check_common(d)
    assert d["config"]["launch"].startswith("torch.autograd.Function")
    assert {"eager", "ext"} <= set(d["launch_modes_ms_per_step"])
    assert d["roofline"]["kernel"].startswith("grid_stage_a_kernel") and d["roofline"]["kernel_ms"] > 0
    assert d["ms_per_step_events_median"] > 0
    assert d["roofline"]["kernel_ms"] < d["fwd_ms"] < d["ms_per_step"]
    for key, metric in (("fps", "fps_point_updates_per_s"), ("ball_group", "group_points_output_bytes_per_s")):
        sub = d[key]
        assert sub["metric"] == metric and sub["value"] > 0 and 0.0 < sub["roofline"]["frac"] < 1.0
        assert "workload" in sub["config"]
    od = d["other_distributions_fwd_ms"]
    assert all(od[k] > 0 for k in ("gaussian", "blobs8", "two_scales", "shapenet_like", "disjoint"))


def test_distributed_line_times_the_exchange_forms(cuda):
    """N > 1 (VERDICT r5 #2c): with no PP_SHARD_EXCHANGE the line times the native all-gather and the grouped
    send / receive form, runs the headline on the faster and reports the step against the wire model.  Two ranks on this
    box's one GPU over gloo (PP_BENCH_DEBUG_GLOO=1: a logic check of the control flow, not a measurement), then one rank
    over RCCL with the p2p form asked for."""
    env = dict(os.environ, PP_BENCH_DEBUG_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PP_SHARD_EXCHANGE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--points", "4096",
                          "--launch", "eager", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert set(d["exchange_modes_ms"]) == {"native", "p2p"} and d["exchange_mode"] in d["exchange_modes_ms"]
    assert all(v > 0 for v in d["exchange_modes_ms"].values())
    m = d["scaling_vs_model"]
    assert m["floor"] == ("all_pairs" if d["exchange_mode"] == "p2p" else "ring") and m["value"] > 0
    assert abs(d["value"] - 2 * 2.0 * 4 * 4096 * 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    env = dict(os.environ, PP_BENCH_FORCE_DIST="1", PP_SHARD_EXCHANGE="p2p", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4",
                          "--points", "4096", "--launch", "eager", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["exchange_mode"] == "p2p" and d["exchange_modes_ms"] is None and "coalesced send / recv" in d["exchange_issue"]
