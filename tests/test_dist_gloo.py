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
