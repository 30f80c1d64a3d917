"""bench.py's bookkeeping without a GPU: the algorithmic bytes of SURVEY.md §8(d) and the choice
of the workload from the free device memory."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_the_survey():
    b = _bench()
    assert (b.b_alg_lbm(4), b.b_alg_step(4)) == (1808, 1856)  # cfg3/4/5
    assert b.b_alg_step(3) == 1416                             # cfg2
    assert b.b_alg_step(1) == 464                              # cfg1: fluid lattice only, no Poisson
    assert b.HBM_PEAK_GBS == 8000.0
    # 60 % of the cfg3 roofline is the target the survey sets: 2 586 MLUPS
    assert round(0.6 * b.HBM_PEAK_GBS * 1e9 / b.b_alg_step(4) / 1e6) == 2586


def test_workload_follows_the_free_memory():
    b = _bench()
    gb = 10**9
    assert b.parse_workload("auto", 300 * gb)[:3] == ("cfg3", (512, 512, 512), 4) and b.parse_workload("auto", 300 * gb)[3] is False
    name, grid, nl, in_place = b.parse_workload("auto", 200 * gb)      # not enough for two buffers
    assert (name, grid, nl, in_place) == ("cfg3", (512, 512, 512), 4, True)
    assert b.parse_workload("auto", 100 * gb)[:3] == ("cfg2", (256, 256, 256), 3)
    assert b.parse_workload("cfg1", 0)[1:3] == ((64, 64, 64), 1)
    assert b.parse_workload("512x512x1024", 0, True) == ("512x512x1024", (512, 512, 1024), 4, True)


def test_plain_invocation_with_gpus_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus 2` invoked plainly (the way the driver invokes --gpus 1) must start
    its two ranks itself - as child processes, before anything touches a GPU - relay rank 0's one
    JSON line and return the child's exit code.  --dry-run stops after the rendezvous and a barrier,
    so this runs without a GPU."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["metric"].startswith("MLUPS")


def test_world_size_must_match_gpus():
    import subprocess
    import sys

    env = dict(os.environ, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
