"""CPU, world_size 2, gloo: the multi-rank logic of bench.py (column-block
partition, barrier, sum/max aggregation) with the oracle standing in for the
GPU step.  Checks that the partition covers every cell once and that the
aggregated counters equal a single-rank run -- i.e. results do not depend on
how cells are split (RNG is keyed by the global cell id)."""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cells_step(cell_ids, levels, n_part):
    from oracle import pmco_py as o
    import bench
    dens = bench.bench_density(1)
    ve, vs = o.make_bin_tables(80, 1e-10, 1e-2)
    temp, pres = bench.level_profile(levels)
    out = {}
    for gid in cell_ids:
        k = gid % levels
        c = o.Cell(1, 1, 0, 256, dens, ve, temp[k], pres[k])
        c.mag[:] = 1e12 / n_part
        c.generate(bench.bench_modes(1), n_part, 0, gid, first_id=gid << 32)
        c.kmax[:] = o.kmax_table(o.KERNEL_BROWN, vs, dens.min(), dens.max(), temp[k], pres[k])
        st, ns, nc = c.mc_coag(o.KERNEL_BROWN, 600.0, 0, gid, 0)
        out[gid] = (ns, nc, int(c.n_part[0]))
    return out


def _worker(rank, world, port, n_cols, levels, n_part, path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c0, c1 = bench.partition_columns(n_cols, world, rank)
    cells = list(range(c0 * levels, c1 * levels))
    dist.barrier()
    res = _cells_step(cells, levels, n_part)
    dist.barrier()
    local = dict(elapsed=0.1 * (rank + 1), n_samp=sum(v[0] for v in res.values()),
                 n_coag=sum(v[1] for v in res.values()),
                 particle_steps=sum(v[2] for v in res.values()), n_cells=len(cells))
    agg = bench.aggregate(local, dist)
    if rank == 0:
        json.dump(agg, open(path, "w"))
    dist.destroy_process_group()


def test_partition_covers_all_columns():
    sys.path.insert(0, ROOT)
    import bench
    for n_cols, world in ((10000, 8), (7, 2), (5, 4), (1, 1), (3, 8)):
        blocks = [bench.partition_columns(n_cols, world, r) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n_cols
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_two_ranks_equal_one_rank(tmp_path, oracle):
    import torch.multiprocessing as mp
    n_cols, levels, n_part = 5, 3, 200
    path = str(tmp_path / "agg.json")
    mp.spawn(_worker, args=(2, _free_port(), n_cols, levels, n_part, path), nprocs=2, join=True)
    agg = json.load(open(path))
    sys.path.insert(0, ROOT)
    serial = _cells_step(range(n_cols * levels), levels, n_part)
    assert agg["n_cells"] == n_cols * levels
    assert agg["n_samp"] == sum(v[0] for v in serial.values())
    assert agg["n_coag"] == sum(v[1] for v in serial.values()) > 0
    assert agg["particle_steps"] == sum(v[2] for v in serial.values())
    assert abs(agg["elapsed"] - 0.2) < 1e-12      # max over ranks


def _grid_worker(rank, world, port, grid, periodic, path):
    """Every rank of a px x py grid sends a tagged message of its own length towards each side; what
    arrives from a side must be the message its neighbour there addressed to the opposite side."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from wrf_partmc_amd import host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nbr = host.grid_neighbours(rank, grid, periodic, periodic)
    out = [torch.full((3 + rank + 2 * side,), 100.0 * rank + side, dtype=torch.float64) for side in range(4)]
    got = host.swap_sides(dist, out, nbr, extent=(3, 3, grid, periodic, periodic))
    ok = True
    # a periodic direction of the whole domain two cells wide is refused, on every rank, before
    # anything is sent (pmcx.h, pmcx_transport_begin): here px blocks of one column each
    try:
        host.swap_sides(dist, out, nbr, extent=(1, 3, grid, periodic, periodic))
        refused = False
    except host.PmcxError:
        refused = True
    ok &= refused == (periodic and grid[0] == 2)
    for side in range(4):
        frm = nbr[side]
        if frm is None:
            ok &= got[side] is None
        else:                                   # the neighbour on my side s sent it towards ITS side s ^ 1
            want = torch.full((3 + frm + 2 * (side ^ 1),), 100.0 * frm + (side ^ 1), dtype=torch.float64)
            ok &= got[side] is not None and got[side].shape == want.shape and bool((got[side] == want).all())
    flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([int(ok)]))
    if rank == 0:
        json.dump([int(f.item()) for f in flags], open(path, "w"))
    dist.destroy_process_group()


@pytest.mark.parametrize("grid,periodic", [((2, 2), True), ((2, 2), False), ((3, 2), True), ((1, 4), False)])
def test_four_neighbour_exchange_of_a_grid_of_ranks(tmp_path, grid, periodic):
    """host.grid_neighbours / host.swap_sides, the exchange pattern of CellBlock.transport_distributed_2d
    and .transport_preweight_distributed_2d (a domain cut in x and y), on CPU over gloo."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from wrf_partmc_amd import host
    px, py = grid
    # a rank's four neighbours, and being each other's opposite
    for r in range(px * py):
        nb = host.grid_neighbours(r, grid, periodic, periodic)
        for side, q in enumerate(nb):
            if q is not None:
                assert host.grid_neighbours(q, grid, periodic, periodic)[side ^ 1] == r
        a, b = r % px, r // px
        assert (nb[0] is None) == (not periodic and b == 0) and (nb[3] is None) == (not periodic and a == px - 1)
    path = str(tmp_path / "ok.json")
    mp.spawn(_grid_worker, args=(px * py, _free_port(), grid, periodic, path), nprocs=px * py, join=True)
    assert json.load(open(path)) == [1] * (px * py)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE): bench.py starts one
    child rank per GPU itself, before it touches a GPU, and passes on rank 0's single line and the
    children's exit code.  Here without a GPU (--rehearse-launch: rendezvous over gloo, partition,
    barrier, aggregation); the same command with the real step runs in tests/test_gpu_fullsize.py."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch",
                        "--nx", "3", "--ny", "2", "--levels", "4"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_joined"] == 2 and line["cells"] == 24
    # a launcher that disagrees with --gpus is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"],
                        env=env2, capture_output=True, text=True, timeout=600)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in r2.stderr
