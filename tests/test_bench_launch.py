"""`python bench.py --gpus N` must start N ranks by itself (no launcher), one JSON line from
rank 0, whole-job numbers.  The GPU kernel cannot run here, so the ranks render through the
`--engine` hook (tests/bench_cpu_engine.py: CPU tensors + gloo); everything else - the launch,
the environment of the ranks, the sample split, the pipelined reduce, the resolve and the JSON
line - is the code the GPU run executes."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--engine", "tests.bench_cpu_engine", "--workload", "cornell_1080p", "--width", "40", "--height", "24",
          "--warmup", "1", "--steps", "2", "--no-cpu-baseline"]


def _run(args, env=None, check=True):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stderr[-2000:]
    return r


def test_gpus_2_starts_two_ranks_and_sums_their_work(tmp_path):
    f2, f1 = str(tmp_path / "two.npy"), str(tmp_path / "one.npy")
    two = _run(["--gpus", "2", "--spp-per-step", "2", "--dump-frame", f2] + COMMON)
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, two.stdout  # ONE JSON line, from rank 0
    j2 = json.loads(lines[0])
    one = _run(["--gpus", "1", "--spp-per-step", "4", "--dump-frame", f1] + COMMON)
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["scaling"] == "weak"
    assert j2["config"]["spp_per_step_per_gpu"] == 2 and j2["config"]["spp_total"] == 8 == j1["config"]["spp_total"]
    # both runs rendered the absolute samples [4, 12) of every pixel: same rays, same frame up to
    # the order of the fp32 additions
    rays2 = j2["value"] * 1e6 * j2["ms_per_step"] * 1e-3 * j2["steps"]
    rays1 = j1["value"] * 1e6 * j1["ms_per_step"] * 1e-3 * j1["steps"]
    assert abs(rays2 - rays1) < 1.0, (rays2, rays1)
    assert abs(j2["rays_per_sample"] - j1["rays_per_sample"]) < 1e-9
    a, b = np.load(f2), np.load(f1)
    assert a.shape == (24, 40, 3) and a.sum() > 0
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    # rank 0's own launches saw half of the rays
    assert abs(j2["roofline"]["rays_per_launch"] * 2 * j2["steps"] / rays2 - 1) < 0.2


def test_a_rank_that_dies_ends_the_run_at_once():
    """Rank 1 exits before the rendezvous; rank 0 would wait there for minutes.  The parent must
    stop the others and fail."""
    import time

    t0 = time.time()
    r = _run(["--gpus", "2", "--spp-per-step", "1"] + COMMON, env={"BENCH_ENGINE_FAIL_RANK": "1"}, check=False)
    assert r.returncode == 1 and "ranks failed" in r.stderr, r.stderr[-1000:]
    assert "(1, 3)" in r.stderr  # rank 1, its exit code
    assert time.time() - t0 < 60.0
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())


def test_a_rank_that_hangs_ends_the_run_inside_the_deadline():
    """Rank 1 never returns from its first render step, so rank 0 parks in the reduce.  The parent
    must kill both when --rank-timeout passes, fail, and say where each rank was."""
    import time

    t0 = time.time()
    r = _run(["--gpus", "2", "--spp-per-step", "1", "--rank-timeout", "25"] + COMMON,
             env={"BENCH_ENGINE_HANG_RANK": "1"}, check=False)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 60.0
    assert "--rank-timeout" in r.stderr and "rank 0:" in r.stderr and "rank 1:" in r.stderr
    assert "reduce of step 0" in r.stderr  # rank 0 waits in the collective
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())


def test_a_hanging_rank_under_torchrun_ends_itself():
    """Under torch.distributed.run there is no parent of ours: every rank carries the deadline."""
    import socket
    import time

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["BENCH_ENGINE_HANG_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--spp-per-step", "1", "--rank-timeout", "25"] + COMMON,
                       env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert time.time() - t0 < 90.0
    assert "did not finish within --rank-timeout" in r.stderr
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--spp-per-step", "1"] + COMMON, env={"WORLD_SIZE": "1", "RANK": "0"}, check=False)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    r = _run(["--gpus", "1", "--spp-per-step", "1"] + COMMON,
             env={"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}, check=False)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_more_gpus_than_visible_is_refused():
    # no --engine: the GPU path; this machine (and a one-GPU box) has fewer than 64 devices
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], check=False)
    assert r.returncode != 0 and "visible" in r.stderr
    assert r.stdout.strip() == ""


def test_under_torchrun_the_script_is_one_of_the_ranks(tmp_path):
    """The scaling run's own command line: torch.distributed.run starts the ranks, bench.py must
    not start more."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    f = str(tmp_path / "frame.npy")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--spp-per-step", "1", "--dump-frame", f] + COMMON,
                       env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["spp_total"] == 4
    assert np.load(f).sum() > 0
