"""The bench contract: one JSON line with the agreed keys, for the four step forms."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["triple", "double", "fused", "sweeps"])
def test_bench_prints_one_json_line(form):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--no-cpu", "--g-grid-per-gpu", "65536"] + ([] if form == "triple" else ["--form", form]),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["ms_per_step"] * 1e-3 * d["value"] - 128 * 65536) < 1e-3 * 128 * 65536
    assert "workload" in d["config"] and d["config"]["form"] == form
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None and rf["traffic_source"] is None   # counter figures on file are for full-size lattices
    assert d["repeats"] == 5 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert d["ranks_seen"] == 1
    if form == "sweeps":
        assert "two_sweep" not in d
    else:
        two = d["two_sweep"]                # the reference's formulation, timed in the same process
        assert two["kernel"] == "sweep_pipe_kernel" and two["steps"] == 10 and two["frac"] > 0
        assert abs(two["frac"] - two["achieved"] / two["peak"]) < 1e-12
    if form == "triple":                    # the default on one GPU: three steps per launch, the double step beside it
        assert rf["kernel"] == "ring_step_kernel" and "note" in rf
        assert d["double_step"]["kernel"] == "fused2_step_kernel" and d["double_step"]["frac"] > 0


@pytest.mark.parametrize("driver", ["auto", "torch"])
def test_bench_over_two_slabs_prints_the_whole_job_rate(driver):
    """Two ranks sharing the GPU (RCCL refuses two ranks on one device, so the C slab driver's
    messages go over sockets, the torch driver's over gloo): rank 0 prints one line, `value`
    counts the columns of both slabs, the line says which driver ran, what each rank took,
    and that the ghost-column check before the timing passed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--g-grid-per-gpu", "65536", "--driver", driver]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["driver_ran"] == ("c" if driver == "auto" else "torch") and d["ranks_seen"] == 2
    assert d["config"]["transport"] == ("fds" if driver == "auto" else "torch.distributed gloo")
    assert d["self_check"]["ok"] is True and d["self_check"]["ranks"] == 2
    assert len(d["ms_per_step_by_rank"]) == 2 and all(0 < x <= d["ms_per_step_max"] * 1.001 for x in d["ms_per_step_by_rank"])
    assert "g-grid=131072" in d["config"]["workload"] and d["config"]["parallelism"] == "phi_y-slab x2"
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 128 * 131072) < 1e-3 * 128 * 131072
    # the C driver takes three steps per launch, the torch driver two
    assert d["roofline"]["kernel"] == ("ring_step_kernel" if driver == "auto" else "fused2_step_kernel") and "note" in d["roofline"]
    assert d["config"]["form"] == ("triple" if driver == "auto" else "double")


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the process starts the two ranks itself
    (before it touches a GPU), their one JSON line comes through, and so does their status."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--repeats", "2", "--g-grid-per-gpu", "65536"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # nothing but that line on standard output, whatever the libraries underneath announce
    assert r.stdout.strip() == lines[0], r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["repeats"] == 2
    assert "g-grid=131072" in d["config"]["workload"]
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 128 * 131072) < 1e-3 * 128 * 131072
    # a failing rank's status comes back too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--g-grid-per-gpu", "4"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode != 0


def test_bench_strong_scaling_cuts_one_lattice_into_slabs():
    """--scaling strong --g-grid G: the whole job is ONE lattice of G columns whatever the
    number of GPUs (how config 4 is run on 2, 4 and 8 of them)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "2",
                        "--repeats", "2", "--scaling", "strong", "--g-grid", "300000"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and "g-grid=300000 " in d["config"]["workload"]
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 128 * 300000) < 1e-3 * 128 * 300000
    assert d["self_check"]["ok"] is True and d["driver_ran"] == "c"


@pytest.mark.parametrize("gpus,steps,warmup", [(1, 7, 1), (1, 8, 3), (1, 2, 0), (2, 5, 3), (2, 1, 0)])
def test_bench_takes_odd_step_counts(gpus, steps, warmup):
    """Whatever --steps and --warmup the caller picks: an odd count ends on a single step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps),
                        "--warmup", str(warmup), "--repeats", "2", "--g-grid-per-gpu", "65536", "--no-cpu",
                        "--no-two-sweep"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == steps and d["warmup"] == warmup and d["value"] > 0
    # the launch time of the dominant kernel comes from blocks of whole launches whatever --steps is
    assert d["roofline"]["kernel"] == "ring_step_kernel" and d["roofline"]["avg_launch_ms"] > 0


def test_bench_stress_data_fills_every_array():
    """--data stress: same contract, the line says what the lattice held."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--repeats", "2",
                        "--g-grid-per-gpu", "65536", "--no-cpu", "--no-two-sweep", "--data", "stress"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip())
    assert "sine products" in d["data"] and d["value"] > 0 and d["roofline"]["kernel"] == "ring_step_kernel"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--data", "stress",
                        "--g-grid-per-gpu", "65536"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
