"""bench.py's bookkeeping without a GPU: the algorithmic bytes of SURVEY.md §8(d) and the choice
of the workload from the free device memory."""
import importlib.util
import json
import os
import subprocess
import sys
import time

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


def test_multi_gpu_lines_are_the_baseline_configs_by_name():
    """`bench.py --gpus N` must run BASELINE.json's configs[3..4]: cfg4 (512x512x1024 split over the ranks) at N = 2
    and 4, cfg5 (1024^3, 128 planes per rank) at N = 8, and say truthfully whether that is weak or strong scaling
    against the N=1 line (cfg3, 134 M nodes on the one GPU)."""
    b = _bench()
    gb = 10**9
    one = b.select_workload("auto", 1, 300 * gb)
    assert one["label"].startswith("cfg3: 512x512x512") and one["scaling"] == "weak" and one["nodes_per_rank"] == 512**3
    two = b.select_workload("auto", 2, 300 * gb)
    assert two["label"].startswith("cfg4: 512x512x1024") and two["grid"] == (512, 512, 1024)
    assert two["planes_per_rank"] == [512, 512] and two["nodes_per_rank"] == 512**3 and two["scaling"] == "weak" and not two["in_place"]
    four = b.select_workload("auto", 4, 300 * gb)
    assert four["label"].startswith("cfg4: 512x512x1024") and four["planes_per_rank"] == [256] * 4
    assert four["nodes_per_rank"] == 512**3 // 2 and four["scaling"] == "strong"
    eight = b.select_workload("auto", 8, 300 * gb)
    assert eight["label"].startswith("cfg5: 1024x1024x1024") and eight["planes_per_rank"] == [128] * 8
    assert eight["nodes_per_rank"] == 512**3 and eight["scaling"] == "weak" and "over 8 GPUs" in eight["label"]
    # cfg4 by name at N = 8: the strong-scaling end, 128 planes of 512x512 per rank
    s8 = b.select_workload("cfg4", 8, 300 * gb)
    assert s8["label"].startswith("cfg4: 512x512x1024") and s8["planes_per_rank"] == [128] * 8 and s8["scaling"] == "strong"
    assert s8["nodes_per_rank"] == 512**3 // 4
    # the 512^3-slab-per-rank channel of rounds 1-2 only behind an explicit --weak
    w4 = b.select_workload("cfg3", 4, 300 * gb, weak=True)
    assert w4["grid"] == (512, 512, 2048) and w4["scaling"] == "weak" and w4["planes_per_rank"] == [512] * 4
    # a rehearsal with fewer planes says so and never passes for the BASELINE size
    r2 = b.select_workload("auto", 2, 300 * gb, scale_z=8)
    assert r2["grid"] == (512, 512, 128) and "REHEARSAL" in r2["label"] and r2["scaling"] == "strong"
    # a rank without room for two population buffers makes every rank run in place (free_bytes = minimum over the ranks)
    assert b.select_workload("auto", 2, 200 * gb)["in_place"] is True
    # uneven slabs are named as a range; too thin slabs and too many ranks are refused
    u = b.select_workload("512x512x513", 8, 300 * gb)
    assert sorted(set(u["planes_per_rank"])) == [64, 65] and "64-65 planes" in u["label"]
    import pytest

    with pytest.raises(ValueError):
        b.select_workload("16x16x12", 4, gb)
    with pytest.raises(ValueError):
        b.select_workload("cfg5", 17, gb)
    # device_need_bytes is what capi.hip allocates for cfg3 (232.8 GB populations + fields)
    assert abs(b.device_need_bytes(512, 512, 512, 4, False) / 1e9 - 250.0) < 2.0
    assert abs(b.device_need_bytes(512, 512, 512, 4, True) / 1e9 - 148.7) < 2.0


def test_comm_block_keys():
    """what an N>1 line must carry to explain its own scaling loss"""
    b = _bench()
    raw = {"halo": {"n": 20, "wait_ms": 4.0, "transfer_ms": 30.0, "bytes_sent": 151_000_000},
           "edge": {"n": 20, "wait_ms": 10.0, "transfer_ms": 9.0, "bytes_sent": 4_200_000},
           "phi": {"n": 20, "wait_ms": 2.0, "transfer_ms": 1.0, "bytes_sent": 4_194_304}}
    c = b.comm_block(raw, 20, 2 * 9 * 4 * 8 * 512 * 512, [0.8, 0.9])
    assert c["halo_bytes_per_step"] == 151_000_000 and c["halo_bytes_per_step_formula"] == 150_994_944
    for k in ("halo", "edge", "phi"):
        assert set(c[k]) == {"exchanges_per_step", "bytes_sent_per_exchange", "wait_ms_per_step", "transfer_ms_per_step"}
        assert c[k]["exchanges_per_step"] == 1.0
    assert c["halo"]["wait_ms_per_step"] == 0.2 and c["edge"]["transfer_ms_per_step"] == 0.45
    assert c["wait_ms_per_step"] == 0.8 and c["transfer_ms_per_step"] == 2.0 and c["wait_ms_per_step_by_rank"] == [0.8, 0.9]


def test_cpu_budget_is_read_not_assumed():
    b = _bench()
    n, how = b.host_cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1) and isinstance(how, str) and how


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
    # the workload a real run would take: cfg4 by name, 512 planes per rank, weak against the N=1 line
    assert out["config"]["workload"].startswith("cfg4: 512x512x1024") and out["config"]["planes_per_rank"] == [512, 512] and out["scaling"] == "weak"


def test_world_size_must_match_gpus():
    import subprocess
    import sys

    env = dict(os.environ, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def _run_dry(args, extra_env, timeout=600):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_headline_multi_gpu_line_fails_closed_without_the_library_transport():
    """the N>1 line runs on the library's own RCCL transport or not at all.  With an RCCL library that
    cannot be bound (EKPNP_RCCL_LIBRARY names a file that does not exist) every rank must leave non-zero with the library's
    message and print NO JSON line - the decision needs no GPU, so --dry-run takes the same code path
    (bench.transport_or_exit).  Only the explicit --allow-fallback-transport lets the example transport stand in, and the
    line then says so."""
    import json

    bad = {"EKPNP_RCCL_LIBRARY": "/nonexistent/librccl.so.1"}
    r = _run_dry(["--gpus", "2", "--dry-run"], bad)
    assert r.returncode != 0, r.stdout
    assert "cannot be loaded" in r.stderr and "the library's RCCL transport cannot be set up" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")], r.stdout
    # one rank through the multi-rank code path: same rule
    r1 = _run_dry(["--force-slab", "--dry-run"], bad)
    assert r1.returncode != 0 and "cannot be loaded" in r1.stderr
    # opt-in: the line is printed and names the transport that would run
    ok = _run_dry(["--gpus", "2", "--dry-run", "--allow-fallback-transport"], bad)
    assert ok.returncode == 0, ok.stderr[-2000:]
    out = json.loads([l for l in ok.stdout.splitlines() if l.strip()][0])
    assert "FALLBACK" in out["transport"]
    # and with RCCL in place the dry run names the library's transport
    good = _run_dry(["--gpus", "2", "--dry-run"], {})
    assert good.returncode == 0, good.stderr[-2000:]
    assert json.loads([l for l in good.stdout.splitlines() if l.strip()][0])["transport"] == "RCCL inside libekpnp.so"


def test_dry_run_rejects_a_bad_workload_on_every_rank():
    """only rank 0 used to select the workload in --dry-run, and its ValueError left the other ranks in a barrier."""
    r = _run_dry(["--gpus", "2", "--dry-run", "--workload", "16x16x6"], {}, timeout=300)
    assert r.returncode != 0 and "bench.py:" in r.stderr and "Traceback" not in r.stderr.split("bench.py:")[0][-400:], r.stderr[-2000:]


def test_the_package_has_no_second_data_path():
    """The python transport is an example outside the package (examples/host_transport.py), not product code."""
    pkg_dir = os.path.join(ROOT, "ek-pnp-3d_amd")
    assert not os.path.exists(os.path.join(pkg_dir, "slab.py"))
    for name in os.listdir(pkg_dir):
        if name.endswith(".py"):
            txt = open(os.path.join(pkg_dir, name)).read()
            for verb in ("P2POp", "isend", "irecv", "all_gather", "all_reduce", "init_process_group", "batch_isend_irecv"):
                assert verb not in txt, (name, verb)  # nothing in the package moves data (or anything else) through torch.distributed
    assert os.path.exists(os.path.join(ROOT, "examples", "host_transport.py"))


def test_ranks_launched_by_torch_distributed_run_get_the_ipc_mode_too():
    """`python -m torch.distributed.run ... bench.py --gpus N` is the DRIVER's launch, and its ranks
    used to start without HSA_ENABLE_IPC_MODE_LEGACY=0 (only bench.py's own spawn_ranks set it) - RCCL across processes
    fails without it on this pool.  bench.py now sets it at import, before torch; the dry run reports what every rank has."""
    import json
    import socket
    import subprocess
    import sys

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert out["env_by_rank"]["HSA_ENABLE_IPC_MODE_LEGACY"] == ["0", "0"]
    # stdout of this launch mode is the ONE line and nothing else: gloo's "[Gloo] Rank r is connected to ..." lines (C-level
    # writes to fd 1 from every rank, found by the GPU test of this launch) go to stderr like RCCL's banner
    assert [l for l in r.stdout.splitlines() if l.strip()] == [json.dumps(out)], r.stdout[-2000:]
    # an explicit setting of the caller is respected, not overwritten
    r = subprocess.run(cmd, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])["env_by_rank"]["HSA_ENABLE_IPC_MODE_LEGACY"] == ["1", "1"]


def test_step_traffic_comes_from_the_committed_counter_table():
    """the line says what the step REALLY moves (all kernels, counter bytes), beside the 1 856 B/node
    yardstick.  profiles/pmc_traffic.json carries the per-kernel table; bench.step_traffic_of sums it."""
    import json

    b = _bench()
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["cfg3"]
    st = rec["step"]
    names = " ".join(e["kernel"] for e in st["kernels"])
    for k in ("k_collide_bulk<4, true, true>", "k_collide_wall<4, true, true>", "k_fft_x_r2c<512>", "k_fft_x_c2r<512>", "k_fft_y512", "k_tridiag_part<8, 64"):
        assert k in names, k
    tot = b.step_traffic_of(rec)
    assert abs(tot - st["hbm_bytes_per_step"]) < 1.0
    alg = b.b_alg_step(4) * 512**3
    # lazy E: the step moves LESS than the yardstick's Poisson I/O (no E arrays) and more for the transform passes: 1.0 - 1.06 x
    assert 1.0 <= tot / alg <= 1.06, tot / alg
    assert b.step_traffic_of({}) is None and b.step_traffic_of({"step": {"kernels": [{"kernel": "k", "launches_per_step": 1, "hbm_bytes_per_launch": None}]}}) is None
    # the keys are in the line bench.py prints
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"step_traffic_bytes"', '"step_hbm_GBps"', '"step_roofline_frac"'):
        assert key in src


def test_comm_ab_leg_keys_and_restoration():
    """the N>1 line decides the knobs one GPU cannot decide - a few steps under each after the timed
    region, through ekpnp_tune on the live context.  Without a GPU: a stand-in context records the calls; the leg must set
    its knobs first, time `steps` steps between barriers, and report the keys the judge reads."""
    b = _bench()
    calls = []

    class Fake:
        def tune(self, k, v):
            calls.append(("tune", k, v))

        def step(self, n):
            calls.append(("step", n))

        def kernel_timing(self, on):
            calls.append(("timing", on))

        def kernel_timing_get(self):
            return 10, 390.0, 1 << 27

        def poisson_stage_timing_get(self):
            return 10, {"stage1": 9.0, "edge_exchange": 4.0, "stage2": 11.0, "phi_exchange": 0.5, "stage3": 0.1}

        def phase_timing_get(self):
            return 10, 24.6

        def comm_timing_get(self):
            return {k: {"n": 10, "wait_ms": w, "transfer_ms": 1.0, "bytes_sent": 1} for k, w in (("halo", 0.1), ("edge", 4.0), ("phi", 0.5))}

    f = Fake()
    leg = b.comm_ab_leg("edge_chunks=4", [("edge_chunks", 4)], f, f, 10, lambda: calls.append(("barrier",)), None, None, 1, {"edge_chunks": 1})
    assert calls[0] == ("tune", "edge_chunks", 4) and ("step", 10) in calls and calls.count(("barrier",)) == 2
    assert calls[-1] == ("tune", "edge_chunks", 1)  # back to what the timed region ran with
    assert calls.index(("timing", True)) < calls.index(("step", 10)) < calls.index(("timing", False))

    class Refuses(Fake):
        def tune(self, k, v):
            calls.append(("tune", k, v))
            if v == 8:
                raise RuntimeError("hipExtStreamCreateWithCUMask: refused")

    calls.clear()
    r = Refuses()
    bad = b.comm_ab_leg("comm_cus=8", [("comm_cus", 8)], r, r, 10, lambda: calls.append(("barrier",)), None, None, 1, {"comm_cus": 0})
    assert bad == {"knob": "comm_cus=8", "error": "hipExtStreamCreateWithCUMask: refused"} and not any(c[0] == "step" for c in calls)
    assert calls[-1] == ("tune", "comm_cus", 0)  # a refused leg costs nothing but itself: no step was taken, the knob is back
    for key in ("knob", "steps", "ms_per_step", "collide_bulk_ms", "poisson_ms", "halo_wait_ms", "edge_wait_ms", "phi_wait_ms",
                "stage1_ms", "edge_exchange_ms", "stage2_ms", "phi_exchange_ms", "stage3_ms"):
        assert key in leg, key
    assert leg["knob"] == "edge_chunks=4" and leg["collide_bulk_ms"] == 39.0 and leg["edge_wait_ms"] == 0.4 and leg["stage2_ms"] == 1.1
    # the legs cover every knob the verdict names, and every knob of a leg has a default to go back to
    knobs = {k for _, ks in b.COMM_AB_LEGS for k, _ in ks}
    assert knobs == {"inline_exchanges", "comm_cus", "lead_planes", "edge_chunks", "edge_p2p"} and knobs <= set(b.AB_DEFAULTS)
    assert b.COMM_AB_LEGS[0] == ("defaults", [])
    assert b.ab_baseline() == {"inline_exchanges": 1, "comm_cus": 0, "lead_planes": 2, "edge_chunks": 1, "edge_p2p": 0}
    mm = b.min_max_by_rank([{"collide_bulk": 38.7, "poisson": 2.1, "rest": 0.3}, {"collide_bulk": 38.9, "poisson": 2.6, "rest": 0.2}])
    assert mm == {"min": {"collide_bulk": 38.7, "poisson": 2.1, "rest": 0.2}, "max": {"collide_bulk": 38.9, "poisson": 2.6, "rest": 0.3}}
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"comm_ab"', '"phases_ms_per_step_by_rank"', '"poisson_stages_ms_per_solve_by_rank"'):
        assert key in src


def test_comm_ab_legs_agree_over_two_gloo_ranks(tmp_path):
    """World size 2 over gloo, launched like the driver launches bench.py: the knob A/B of the N>1 line must never leave one rank
    stepping alone.  Rank 1 refuses `comm_cus=8`: BOTH ranks report that leg (and the combined one) as refused and take no step
    for it; the other legs carry the maxima over the ranks; every knob goes back to its baseline."""
    import json
    import socket
    import subprocess
    import sys

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = tmp_path / "ab.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EKPNP_AB_OUT=str(out), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_comm_ab_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = json.load(open(out))
    assert [x["rank"] for x in ranks] == [0, 1]
    for x in ranks:
        legs = {l["knob"]: l for l in x["legs"]}
        assert set(legs) == {"defaults", "inline_exchanges=0", "comm_cus=8", "lead_planes=0", "edge_chunks=4", "edge_chunks=4 comm_cus=8", "edge_p2p=1"}
        assert "error" in legs["comm_cus=8"] and "error" in legs["edge_chunks=4 comm_cus=8"]
        assert ("refused on another rank" in legs["comm_cus=8"]["error"]) == (x["rank"] == 0)
        for k in ("defaults", "inline_exchanges=0", "lead_planes=0", "edge_chunks=4", "edge_p2p=1"):
            # maxima over the ranks: rank 1's sweep (200 ms / 10 steps), waits (2 ms / 10) and EDGE stage (4 ms / 10 solves)
            assert legs[k]["collide_bulk_ms"] == 20.0 and legs[k]["halo_wait_ms"] == 0.2 and legs[k]["edge_exchange_ms"] == 0.4, legs[k]
        steps = [c for c in x["calls"] if c[0] == "step"]
        assert len(steps) == 2 * 5  # (2 settling + 10 timed) for the five legs that ran, none for the refused ones
        tunes = [c for c in x["calls"] if c[0] == "tune"]
        final = {}
        for _, k, v in tunes:
            final[k] = v
        assert final == {"inline_exchanges": 1, "comm_cus": 0, "lead_planes": 2, "edge_chunks": 1, "edge_p2p": 0}


def test_headline_guard_prints_the_measured_line_when_a_later_leg_hangs(tmp_path):
    """What runs after the timed region (knob A/Bs, copy probe, CPU baseline) must not be able to lose the measured line:
    a leg that never returns ends in the line as it stood, marked, and exit code 0 - in a real process, through os._exit."""
    script = tmp_path / "hang.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "g = bench.HeadlineGuard(0, 0.5)\n"
        "g.arm({'metric': 'MLUPS (full EK-PNP step)', 'value': 3268.0, 'config': {'batch_moments_ab': None}})\n"
        "g.update({'metric': 'MLUPS (full EK-PNP step)', 'value': 3268.0, 'config': {'batch_moments_ab': {'x': 1}}}, leg='comm_ab: comm_cus=8')\n"
        "time.sleep(60)\n"
        "print('never')\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "never" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 3268.0 and d["config"]["batch_moments_ab"] == {"x": 1}  # the line as of the last finished leg
    assert d["after_the_fact"]["status"] == "abandoned" and d["after_the_fact"]["leg"] == "comm_ab: comm_cus=8"
    assert "comm_ab: comm_cus=8" in r.stderr


def test_headline_guard_other_ranks_leave_quietly_and_finish_wins_the_race():
    b = _bench()
    rd, wr = os.pipe()
    left = []
    # a rank without the line (rank > 0): nothing on stdout, it just leaves with 0
    g = b.HeadlineGuard(3, 0.05, fd=wr, exit_fn=left.append)
    g.arm(None)
    time.sleep(0.5)
    assert left == [0]
    assert g.finish({"late": True}) is False  # the timer won: nothing more is written
    # the normal end: the complete line, once, and a timer that never fires afterwards
    g2 = b.HeadlineGuard(0, 0.3, fd=wr, exit_fn=left.append)
    g2.arm({"value": 1.0})
    assert g2.finish({"value": 1.0, "cpu_baseline": {"value": 4.2}}) is True
    time.sleep(0.6)
    assert left == [0]
    os.close(wr)
    got = os.read(rd, 1 << 16).decode().splitlines()
    os.close(rd)
    assert [json.loads(x) for x in got] == [{"value": 1.0, "cpu_baseline": {"value": 4.2}}]
    # deadline 0: never armed
    g3 = b.HeadlineGuard(0, 0.0, exit_fn=left.append)
    g3.arm({"value": 2.0})
    assert g3._timer is None


def test_more_ranks_than_devices_ends_every_rank_with_a_message():
    """`bench.py --gpus 2` under the driver's launcher on a box with fewer HIP devices than ranks (here: none): every rank
    leaves together, non-zero, with one sentence that names --single-device - no traceback from torch.cuda.set_device, no rank
    left waiting in a collective, nothing on stdout."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # (a GPU box that runs this file too: no device for anybody)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    # (the two ranks write to one stderr at the same moment: count the sentences, not the lines)
    assert r.stderr.count("0 HIP device(s) visible for 2 ranks") == 2 and r.stderr.count("--single-device for a functional rehearsal") == 2, r.stderr[-3000:]
    assert "Traceback" not in r.stderr.split("failed (exitcode")[0]  # (the launcher prints its own ChildFailedError afterwards)
    assert r.stdout.strip() == ""
