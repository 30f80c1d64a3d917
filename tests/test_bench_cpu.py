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
