"""CPU coverage of the N>1 path: the transport (gloo, world_size 2 and 3, launched exactly like
the driver launches bench.py) and the slab partition rule."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_ring_transport_gloo(world, tmp_path):
    ok = tmp_path / "ok"
    env = dict(os.environ, EKPNP_RING_OK=str(ok), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_ring_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert ok.read_text() == f"ok {world}"


def test_slab_extent_rules(pkg):
    from ek_pnp_3d_amd import slab_extent

    assert slab_extent(512, 0, 1) == (0, 512)
    assert [slab_extent(1024, r, 8) for r in (0, 3, 7)] == [(0, 128), (384, 128), (896, 128)]
    # the reference's usual NZ = 2^k + 1: slabs differ by at most one plane and tile the channel
    assert [slab_extent(51, r, 2) for r in (0, 1)] == [(0, 25), (25, 26)]
    ext = [slab_extent(513, r, 8) for r in range(8)]
    assert ext[0][0] == 0 and ext[-1][0] + ext[-1][1] == 513
    assert all(a[0] + a[1] == b[0] for a, b in zip(ext, ext[1:])) and {e[1] for e in ext} == {64, 65}
    with pytest.raises(ValueError):
        slab_extent(12, 0, 4)
