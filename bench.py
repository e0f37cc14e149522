#!/usr/bin/env python3
"""Throughput of the PartMC per-cell microphysics step on MI355X.

Workload (BASELINE.json configs[4], the configuration the metric is quoted
on): synthetic 100 x 100 x 30 domain, 10^4 computational particles per cell,
Brownian coagulation (the kernel the reference selects,
interface/wrf_pmc_driver.F90:192) with del_t = 300 s (partmc_chem_dt of the
reference's only coagulation-enabled namelist), followed by
aero_state_rebalance (driver:248-250).  One "step" = one pass of that loop
body over every cell.  Cells are partitioned by contiguous blocks of (i,j)
columns over the ranks (one process per GPU, no data-path collective);
total work is fixed as N grows ("strong").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The first form starts its own ranks when N > 1 and no launcher set WORLD_SIZE: a child
`python -m torch.distributed.run` of this file, started before this process touches the GPU;
rank 0's line and the children's exit code are passed on.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# Random-gather ceiling of the chip, measured with scripts/gather_probe.hip (profiles/
# r1f_gather_probe.txt): independent 8..16-byte loads at random addresses of a 1..160 GiB buffer
# complete at 53e9 per second whatever the footprint or the loads in flight per lane; the
# memory side moves 64 B for each (FETCH_SIZE), i.e. 3.4 TB/s of its 8 TB/s.
GATHER_PEAK_PER_S = 53e9

KERNELS = ("bin_sort", "coag_sample", "compact", "rebalance", "condense")


def partition_columns(n_cols, world, rank):
    """Contiguous block of (i,j) columns owned by `rank` (all levels of a column
    stay together, as in WRF's horizontal decomposition,
    WRFV3/main/module_wrf_top.F:505-510)."""
    return rank * n_cols // world, (rank + 1) * n_cols // world


def level_profile(levels):
    """Standard-atmosphere T/p by model level (synthetic env_state)."""
    z = np.linspace(50.0, 12000.0, levels)
    temp = 288.15 - 6.5e-3 * z
    pres = 101325.0 * (temp / 288.15) ** 5.2559
    return temp, pres


def algorithmic_bytes(name, n_spec, n_class, n_source, particles, candidates, events):
    """Algorithmic HBM bytes of one launch: SURVEY.md 8(d)'s per-unit figures, nothing added."""
    meta = 28  # id 8 + create times 16 + class 4
    if name == "bin_sort":
        return particles * (8 + 4)
    if name == "coag_sample":
        # per candidate pair 2 x (cached volume 8 + cached density 8) = 32 B; per
        # accepted event read 2 x n_spec x 8 + write n_spec x 8, 2 x 28 B of metadata read and
        # 28 B written: 24 n_spec + 84 (108 B with one species); the source histogram's counts
        # (4 B per source, 2 read + 1 written) where the run carries sources
        per_ev = 3 * 8 * n_spec + 3 * 4 * n_source + 3 * meta
        return candidates * 32 + events * per_ev
    if name == "condense":
        # read the species (+ class), write water volume and the two caches
        return particles * (8 * n_spec + 4 + 8 + 16)
    return 0


def aggregate(local, dist=None):
    """Sum counters, max time over ranks (the contract's max-over-ranks timing)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    import torch
    keys_sum = [k for k in local if k != "elapsed"]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(local[k]) for k in keys_sum], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    e = torch.tensor([float(local["elapsed"])], dtype=torch.float64, device=dev)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(keys_sum, t.tolist())}
    out["elapsed"] = float(e.item())
    return out


def host_cpu_info():
    """CPU model, sockets and logical cores of this host (from /proc/cpuinfo), and the cores
    this process may run on."""
    model, sockets, logical = "unknown", set(), 0
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                sockets.add(ln.split(":", 1)[1].strip())
            elif ln.startswith("processor"):
                logical += 1
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    # the container's CPU quota (cgroup v2 cpu.max "quota period", v1 cfs_quota_us / cfs_period_us):
    # more runnable processes than this only share the same cores
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        allowed = max(1, min(allowed, int(quota)))
    return dict(model=model, sockets=max(len(sockets), 1), logical_cores=logical or (os.cpu_count() or 1),
                allowed_cores=allowed, cpu_quota=quota)


_BARRIER = None


def _pool_init(barrier):
    global _BARRIER
    _BARRIER = barrier


def cpu_baseline(args, n_cells_total):
    """Times the CPU oracle (a restatement: the reference's PartMC submodule is not vendored,
    BASELINE.md section 2; kind "port") on a bounded sample of the same workload -- cells of the
    same domain, same seeds -- compiled -O3 -march=native on this host: one process
    per allowed core, every worker sets its cells up, all wait at a barrier, then step together
    for at least args.cpu_seconds; plus the single-thread figure.  Also steps the cross-check
    cells (spread over the domain) through every step the GPU will run."""
    import multiprocessing as mp
    from oracle import pmco_py
    pmco_py.build()
    info = host_cpu_info()
    flags = "gcc -O3 -march=native"
    try:
        os.environ["PMCO_LIBRARY"] = pmco_py.build_native()
    except Exception:                                   # no compiler on the box: the portable build
        os.environ.pop("PMCO_LIBRARY", None)
        flags = "gcc -O2 (native build failed)"
    cores = max(1, min(info["allowed_cores"], args.cpu_cores or info["allowed_cores"]))
    ctx = mp.get_context("fork")
    # calibrate in a child (the parent never loads the library: its workers fork clean)
    max_steps = 24          # the populations stay in the regime of the GPU's run (25 steps)
    with ctx.Pool(1, initializer=_pool_init, initargs=(None,)) as pool:
        cal = pool.apply(_oracle_cells, ((args, list(range(64)), 4),))
        per_cell_setup = cal["setup_s"] / 64
        per_cell_step = cal["step_s"] / max(cal["cell_steps"], 1)
        # the single-thread figure: about half of cpu_seconds of stepping, 8 steps
        n1 = int(max(64, min(n_cells_total, 0.5 * args.cpu_seconds / (8 * max(per_cell_step, 1e-9)))))
        one = pool.apply(_oracle_cells, ((args, list(range(n1)), 8),))
    single = one["events"] / max(one["step_s"], 1e-9)
    # cells per worker: set-up bounded to about cpu_setup_seconds; steps: enough for cpu_seconds
    per_worker = int(max(8, min(n_cells_total // cores, args.cpu_setup_seconds / max(per_cell_setup, 1e-6))))
    steps = int(min(max_steps, max(2, np.ceil(args.cpu_seconds / max(per_worker * per_cell_step, 1e-9)))))
    n_sample = per_worker * cores
    chunks = [list(range(k, n_sample, cores)) for k in range(cores)]
    barrier = ctx.Barrier(cores)
    with ctx.Pool(cores, initializer=_pool_init, initargs=(barrier,)) as pool:
        res = pool.map(_oracle_cells, [(args, ch, steps) for ch in chunks], chunksize=1)
    step_wall = max(x["step_s"] for x in res)           # all workers start stepping together
    events = sum(x["events"] for x in res)
    parts = sum(x["particles"] for x in res)
    # the cross-check cells: spread over the domain, stepped through the GPU's whole run
    chk_ids = sorted(set(np.linspace(0, n_cells_total - 1, args.check_cells).astype(int).tolist()))
    n_run = args.warmup + args.steps
    with ctx.Pool(min(cores, 8), initializer=_pool_init, initargs=(None,)) as pool:
        parts_chk = pool.map(_oracle_check, [(args, chk_ids[k::8], n_run) for k in range(8)], chunksize=1)
    chk = {}
    for d in parts_chk:
        chk.update(d)
    out = {
        "value": events / step_wall, "unit": "events/s", "cores": cores, "kind": "port",
        "sample": f"{n_sample} cells x {args.particles} particles x {steps} steps after one warm-up "
                  f"step, {cores} processes stepping together from a barrier (oracle/pmco.c, {flags}; "
                  "set-up and warm-up excluded)",
        "cpu_model": info["model"], "sockets": info["sockets"], "logical_cores": info["logical_cores"],
        "allowed_cores": info["allowed_cores"], "cpu_quota_cores": info["cpu_quota"],
        "particle_steps_per_sec": parts / step_wall,
        "single_thread_events_per_sec": single,
        "single_thread_sample": f"{n1} cells x 8 steps, {one['step_s']:.2f} s",
        "events": events, "seconds": step_wall,
    }
    return out, chk


def _make_oracle_cell(o, args, gid, dens, ve, vs, temp, pres, modes, kcache):
    k = gid % args.levels
    c = o.Cell(args.n_spec, 1, 0, args.cap, dens, ve, temp[k], pres[k],
               max_events=args.max_events, max_appends=args.max_appends)
    c.mag[:] = args.num_conc / args.particles
    c.generate(modes, args.particles, args.seed, gid, first_id=gid << 32)
    if k not in kcache:
        kcache[k] = o.kmax_table(o.KERNEL_BROWN, vs, dens.min(), dens.max(), temp[k], pres[k])
    c.kmax[:] = kcache[k]
    return c


def _oracle_cells(job):
    """Timing worker: set up the cells, one warm-up step (which sorts every cell), then `steps`
    timed steps over all of them."""
    args, cell_ids, steps = job
    from oracle import pmco_py as o
    t_setup = time.perf_counter()
    dens = bench_density(args.n_spec)
    ve, vs = o.make_bin_tables(80, 1e-10, 1e-2)
    temp, pres = level_profile(args.levels)
    modes = bench_modes(args.n_spec)
    kcache = {}
    cells = [(gid, _make_oracle_cell(o, args, gid, dens, ve, vs, temp, pres, modes, kcache))
             for gid in cell_ids]
    for gid, c in cells:
        c.mc_coag(o.KERNEL_BROWN, args.del_t, args.seed, gid, 0)
    t_setup = time.perf_counter() - t_setup
    if _BARRIER is not None:
        _BARRIER.wait(timeout=900)
    ev, parts, cell_steps = 0, 0, 0
    t0 = time.perf_counter()
    for step in range(1, 1 + steps):
        for gid, c in cells:
            parts += int(c.n_part[0])
            st, ns, nc = c.mc_coag(o.KERNEL_BROWN, args.del_t, args.seed, gid, step)
            ev += nc
        cell_steps += len(cells)
    return dict(events=ev, particles=parts, cell_steps=cell_steps,
                step_s=time.perf_counter() - t0, setup_s=t_setup)


def _oracle_check(job):
    """Cross-check worker: the cells through n_run steps of mc_coag + rebalance, as one_step()
    does on the GPU; returns per cell (n_coag of the last step, n_part after it)."""
    args, cell_ids, n_run = job
    from oracle import pmco_py as o
    dens = bench_density(args.n_spec)
    ve, vs = o.make_bin_tables(80, 1e-10, 1e-2)
    temp, pres = level_profile(args.levels)
    modes = bench_modes(args.n_spec)
    kcache, out = {}, {}
    ideal = np.full(1, float(args.particles))
    for gid in cell_ids:
        c = _make_oracle_cell(o, args, gid, dens, ve, vs, temp, pres, modes, kcache)
        nid, nc = int((gid << 32) + (1 << 24)), 0
        for step in range(n_run):
            st, ns, nc = c.mc_coag(o.KERNEL_BROWN, args.del_t, args.seed, gid, step)
            _, nid = c.rebalance(ideal, True, True, args.seed, gid, step, nid)
        out[gid] = (int(nc), int(c.n_part[0]))
    return out


def bench_density(n_spec):
    from wrf_partmc_amd.data import AERO_DATA_SCM
    return np.array([r[1] for r in AERO_DATA_SCM][:n_spec])


def bench_modes(n_spec):
    """Idealised reference IC (interface/wrf_pmc_init.F90:1764-1777): log-normal,
    char_radius 2e-8 m, log10 sigma 0.25, species 1 only."""
    from wrf_partmc_amd.data import idealised_modes
    return idealised_modes(n_spec, 1)


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: one rank per GPU as children of this process
    (`python -m torch.distributed.run` of this file with the same arguments), started BEFORE this
    process has made any GPU call -- a fresh child, never an exec of a process that holds the GPU.
    The children write to this process's stdout / stderr (rank 0 prints the line); returns their
    exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rehearse_launch(args, rank, world):
    """--rehearse-launch: the launch path alone, for a box without a GPU (tests/test_dist_gloo.py) --
    rendezvous over gloo, the column partition, the barrier and the sum / max aggregation, one line
    from rank 0.  No step runs and nothing is measured: `value` is null."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    c0, c1 = partition_columns(args.nx * args.ny, world, rank)
    local = dict(elapsed=0.0, cells=(c1 - c0) * args.levels, ranks=1)
    if world > 1:
        dist.barrier()
    agg = aggregate(local, dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "coagulation_events_per_sec", "value": None, "unit": "events/s",
                          "n_gpus": world, "steps": 0, "warmup": 0, "rehearsal": "launch path only, no GPU",
                          "cells": int(agg["cells"]), "ranks_joined": int(agg["ranks"])}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=100)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--levels", type=int, default=30)
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--n-spec", type=int, default=1)
    ap.add_argument("--num-conc", type=float, default=1e10)   # wrf_pmc_init.F90:1835-1836
    ap.add_argument("--del-t", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)            # registry.partmc:13
    ap.add_argument("--max-events", type=int, default=1024)
    ap.add_argument("--max-appends", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=3.0,
                    help="timed stepping per CPU worker (after set-up, from a common barrier)")
    ap.add_argument("--cpu-setup-seconds", type=float, default=9.0,
                    help="set-up time per CPU worker that bounds its number of cells")
    ap.add_argument("--cpu-cores", type=int, default=0, help="0: every core this process may use")
    ap.add_argument("--check-cells", type=int, default=96,
                    help="cells spread over the domain that the oracle steps through the whole run")
    ap.add_argument("--resort-steps", type=int, default=3,
                    help="extra steps after the timed region with the sort invalidated before "
                         "each (reported as resort_every_step; 0 to skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sampler-lanes", type=int, default=0,
                    help="diagnostic: pin the pair sampler's variant (pmcx_set_sampler_variant); 0 = automatic")
    ap.add_argument("--sampler-wpb", type=int, default=0,
                    help="diagnostic: wavefronts per SIMD the lane-per-cell sampler may hold (LDS padding)")
    ap.add_argument("--sampler-stamps", action="store_true",
                    help="diagnostic: the instantiation that records cycles per stage")
    ap.add_argument("--kernel", type=int, default=4,
                    help="coag_kernel_type (4 = COAG_KERNEL_TYPE_BROWN, the reference's; others for tuning)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N > 1 run: nccl (RCCL, one rank per GPU; what the "
                         "driver's scaling runs use) or gloo (rehearsal: several ranks may share one GPU)")
    ap.add_argument("--workload", choices=["domain", "config3"], default="domain",
                    help="domain: BASELINE configs[4] (default). config3: 1000 cells x 10^4, "
                         "20 species, condensation sub-step + coagulation (BASELINE configs[2])")
    ap.add_argument("--n-class", type=int, default=1,
                    help="weight classes (WRFV3/Registry/registry.partmc_trans:2-41 carries 40): the idealised mode "
                         "split evenly over them, class c standing for (1 + c / 2) times class 0's number "
                         "(a side measurement: no CPU leg; the screen + list-sweep form of the sampler, DESIGN.md 5.3)")
    ap.add_argument("--cap-factor", type=float, default=1.0,
                    help="slots per cell / particles per cell (1: the headline's; the Fortran shim's default headroom is 3; "
                         "with tens of classes the lists' layout needs the room, DESIGN.md 4)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU needed: launch, rendezvous (gloo), partition and aggregation only; value null")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))       # (nothing has touched the GPU yet)
    args.cap = (int(args.particles * args.cap_factor) + 63) // 64 * 64
    config3 = args.workload == "config3"
    if config3:
        args.nx, args.ny, args.levels, args.n_spec = 10, 10, 10, 20
        args.no_cpu_baseline = True
    if args.n_class > 1:
        args.no_cpu_baseline = True

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: one rank per GPU")
    if args.rehearse_launch:
        return rehearse_launch(args, rank, world)
    # CPU baseline first, on rank 0 at N=1 only, before this process touches the GPU
    # (the worker pool forks; a forked child must not inherit a live HIP runtime)
    cpu_line, cpu_counts = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_line, cpu_counts = cpu_baseline(args, args.nx * args.ny * args.levels)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and there is no CPU path")
    # --backend gloo lets several ranks share one GPU (rehearsal of the N>1
    # path on a one-GPU box); the driver's multi-GPU runs use the default, RCCL.
    backend = args.backend
    device = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import wrf_partmc_amd as wp
    n_cols = args.nx * args.ny
    c0, c1 = partition_columns(n_cols, world, rank)
    first_cell, n_cells = c0 * args.levels, (c1 - c0) * args.levels
    n_cells_total = n_cols * args.levels

    blk = wp.CellBlock(n_cells, args.cap, args.n_spec, n_class=args.n_class, n_source=0, device=device,
                       cell_id_base=first_cell, max_events=args.max_events,
                       max_appends=args.max_appends, i_water=(19 if config3 else -1))
    if args.sampler_lanes or args.sampler_stamps or args.sampler_wpb:
        blk.set_sampler_variant(args.sampler_lanes, args.sampler_wpb, int(args.sampler_stamps))
    temp, pres = level_profile(args.levels)
    lev = (first_cell + np.arange(n_cells)) % args.levels
    if config3:
        from wrf_partmc_amd.data import aero_data_arrays
        dens20, kappa20, _, _ = aero_data_arrays()
        blk.set_aero_data(dens20, kappa=kappa20)
        rh = np.random.RandomState(args.seed).uniform(0.3, 0.95, n_cells_total)
        blk.set_env(temp[lev], pres[lev], rh[first_cell:first_cell + n_cells])
    else:
        blk.set_aero_data(bench_density(args.n_spec))
        blk.set_env(temp[lev], pres[lev], np.full(n_cells, 0.5))
    Cn = args.n_class
    # (class c stands for (1 + c / 2) times class 0's number; the classes hold equal shares of the particles)
    w_cls = 1.0 + 0.5 * np.arange(Cn)
    mag0 = args.num_conc / (args.particles / Cn * w_cls.sum())
    blk.set_weights(np.tile(mag0 * w_cls, (n_cells, 1)))
    blk.set_n_part_ideal(np.full((n_cells, Cn), float(args.particles) / Cn))
    blk.set_next_id((first_cell + np.arange(n_cells, dtype=np.int64) << 32) + (1 << 24))
    if config3:
        from wrf_partmc_amd.data import scm_init_modes
        modes3, _ = scm_init_modes()      # em_scm_xy bimodal IC, OC:SO4:NH4 composition
        blk.generate_lognormal(args.particles, modes3, seed=args.seed)
    elif Cn > 1:
        base = bench_modes(args.n_spec)[0]
        blk.generate_lognormal(args.particles, [dict(base, frac=1.0 / Cn, wclass=c) for c in range(Cn)], seed=args.seed)
    else:
        blk.generate_lognormal(args.particles, bench_modes(args.n_spec), seed=args.seed)
    blk.update_kernel_bounds(args.kernel)
    blk.synchronize()

    def barrier():
        blk.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(step):
        if config3:
            blk.condense(args.del_t, 4)   # extension: water uptake sub-step before coagulation
        blk.mc_coag(args.kernel, args.del_t, args.seed, step, counters=False)
        blk.accumulate_totals()      # n_samp, n_coag, n_part, status: summed on the device,
        blk.aero_state_rebalance(True, True, args.seed, step)   # no host sync inside the step

    prof_main = {}
    step = 0
    for _ in range(args.warmup):
        one_step(step)
        step += 1
    blk.read_accumulated(reset=True)      # drop the warm-up steps' sums
    blk.profile_enable(True)
    blk.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(step)
        step += 1
    barrier()
    elapsed = time.perf_counter() - t0
    blk.profile_enable(False)
    # the last timed step's per-cell event counts and the populations it left, for the
    # cross-check against the oracle on cells spread over the whole domain
    last_counts = None
    if cpu_counts is not None:
        _, nc_last, _ = blk.coag_counters()
        np_last = blk.aero_state_total_particles()
        last_counts = {g: (int(nc_last[g - first_cell]), int(np_last[g - first_cell]))
                       for g in cpu_counts if first_cell <= g < first_cell + n_cells}
    sampler_left = blk.sampler_info()     # cells the sampler's split-off draw left to its second launch (last step)
    samp, coag, parts_after, status = blk.read_accumulated(reset=True)
    parts = parts_after + coag    # one class: particles entering a step = survivors + events (several: an upper bound)
    for k in KERNELS:
        prof_main[k] = blk.profile_get(k)
    if args.sampler_stamps and rank == 0:   # diagnostic build of the sampler
        ph = blk.profile_phase_cycles()
        w = max(ph[11], 1)
        names = ["setup", "1_slots_settled_blocks_fetched", "2_exact_counts", "3_pair_picked_entries_requested",
                 "4_records_requested", "5_kernel_accept", "6_unused", "7_coagulation", "write_back"]
        d = {n: ph[k] / w for k, n in enumerate(names)}
        d.update(draws=ph[9] / w, passes=ph[10] / w, wavefronts=ph[11])
        print("seq sampler, mean s_memtime ticks per wavefront: " + json.dumps(d), file=sys.stderr)

    # Side measurement, outside the timed region: the same step when something other than
    # mc_coag touched every cell in between (transport, emission: aero_state%valid_sort
    # false), so that every cell is sorted again on every step.
    resort = None
    if args.resort_steps > 0:
        blk.invalidate_sort()        # warm-up: the keys are computed from the records once
        one_step(step)
        step += 1
        blk.read_accumulated(reset=True)
        blk.profile_enable(True)
        blk.profile_reset()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.resort_steps):
            blk.invalidate_lists()   # (the slots' sort keys hold: what transport or deposition leave)
            one_step(step)
            step += 1
        barrier()
        el_r = time.perf_counter() - t1
        blk.profile_enable(False)
        _, coag_r, _, _ = blk.read_accumulated(reset=True)
        resort = dict(elapsed=el_r, n_coag=coag_r, ms_sort=blk.profile_get("bin_sort")[0])

    local = dict(elapsed=elapsed, n_samp=samp, n_coag=coag, particle_steps=parts)
    for k in KERNELS:
        ms, n = prof_main[k]
        local["ms_" + k] = ms
        local["n_" + k] = n
    if resort:
        local.update(r_n_coag=resort["n_coag"], r_ms_sort=resort["ms_sort"])
    agg = aggregate(local, dist if world > 1 else None)
    if resort and world > 1:
        import torch as _t
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        e = _t.tensor([resort["elapsed"]], dtype=_t.float64, device=dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        resort["elapsed"] = float(e.item())
    mem = blk.memory_bytes()

    if rank == 0:
        el = agg["elapsed"]
        kern_ms = {k: agg["ms_" + k] / world for k in KERNELS}          # mean over ranks
        dom = max(kern_ms, key=kern_ms.get)
        launches = max(1.0, agg["n_" + dom] / world)
        # units one launch of the dominant kernel processes on one GPU, averaged
        per_launch = dict(particles=agg["particle_steps"] / world / args.steps,
                          candidates=agg["n_samp"] / world / args.steps,
                          events=agg["n_coag"] / world / args.steps)
        alg = algorithmic_bytes(dom, args.n_spec, args.n_class, 0, **per_launch)
        avg_ms = kern_ms[dom] / launches
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes (profiles/pmc_traffic.json, scripts/gpu_pmc_shape.sh):
        # taken at a run shape (--warmup, --steps) because the cells thin out as a run goes on; this
        # run's shape when it was profiled, else the nearest profiled shape scaled by events per launch
        # (the sampler's traffic is per event and candidate), and said so
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        full_default = (args.workload == "domain" and (args.nx, args.ny, args.levels) == (100, 100, 30)
                        and args.particles == 10000 and args.n_spec == 1 and world == 1 and args.n_class == 1)
        if full_default and dom == "coag_sample" and os.path.exists(tpath):
            try:
                shapes = json.load(open(tpath)).get("shapes", {})
                key = f"w{args.warmup}s{args.steps}"
                if key in shapes:
                    e = shapes[key]
                    # (the same steps of the same simulation: events per launch agree exactly)
                    traffic = e["bytes_per_launch"] * per_launch["events"] / e["events_per_launch"]
                    traffic_src = f"PMC passes of this run shape ({key}), profiles/pmc_traffic.json"
                elif shapes:
                    k2, e = min(shapes.items(), key=lambda kv: abs(kv[1]["events_per_launch"] - per_launch["events"]))
                    traffic = e["bytes_per_event"] * per_launch["events"]
                    traffic_src = (f"PMC passes of run shape {k2} ({e['events_per_launch']:.4g} events per launch) "
                                   f"scaled by events per launch to this run's {per_launch['events']:.4g}")
            except Exception:
                traffic = None
        line = {
            "metric": "coagulation_events_per_sec", "value": agg["n_coag"] / el,
            "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{args.nx}x{args.ny}x{args.levels} cells x {args.particles} "
                                    f"particles/cell, Brownian mc_coag del_t={args.del_t:g}s + rebalance, "
                                    f"n_spec={args.n_spec}" + (f", n_class={args.n_class}" if args.n_class > 1 else "")
                                    + f", cell-column blocks over {world} GPU(s)"
                                    + (", condensation sub-step (extension) before coagulation, "
                                       "em_scm_xy species table and bimodal IC" if config3 else "")),
                       "cells": n_cells_total, "particles_per_cell": args.particles,
                       "n_spec": args.n_spec, "kernel": "COAG_KERNEL_TYPE_BROWN"},
            "particle_steps_per_sec": agg["particle_steps"] / el,
            "candidate_pairs_per_sec": agg["n_samp"] / el,
            "events_per_step": agg["n_coag"] / args.steps,
            # the populations thin by about 0.5 % a step (no doubling until a cell falls below half
            # its ideal count): ms_per_step depends on which steps are timed, events/s much less
            "n_part_mean": agg["particle_steps"] / args.steps / n_cells_total,
            "timed_steps": f"simulation steps {args.warmup} .. {args.warmup + args.steps - 1}",
            "status_or": status,
            "sampler_cells_left_to_second_launch": sampler_left,
            "device_memory_gb_rank0": mem / 1e9,
            "kernel_ms_per_step": {k: kern_ms[k] / args.steps for k in KERNELS},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms},
        }
        if traffic:
            # the sampler reads single words of randomly picked particles: what bounds it is
            # the rate of 64-byte memory requests, not bytes of payload
            req_s = traffic / 64.0 / (avg_ms * 1e-3)
            line["gather_roofline"] = {
                "kernel": dom, "unit": "64-B requests/s", "achieved": req_s,
                "peak": GATHER_PEAK_PER_S, "frac": req_s / GATHER_PEAK_PER_S,
                "requests_per_launch": traffic / 64.0,
                "source": "FETCH_SIZE + WRITE_SIZE (roofline.traffic_source) over 64 B; "
                          "peak from scripts/gather_probe.hip"}
        # The other kernels of the step.  Only a kernel whose launch streams the whole population
        # gets a bandwidth figure (condense); inside the timed region bin_sort, compact and rebalance
        # run on the few listed cells that need them, so bytes of the whole domain over their
        # duration would be a meaningless number: their time is reported, nothing else (the
        # sort of every cell is measured under resort_every_step).
        others = {}
        for k in KERNELS:
            if k == dom or agg["n_" + k] == 0:
                continue
            ms = kern_ms[k] / max(1.0, agg["n_" + k] / world)
            others[k] = {"avg_launch_ms": ms}
            if k == "condense":
                a = algorithmic_bytes(k, args.n_spec, 1, 0, **per_launch)
                others[k]["achieved"] = a / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        line["roofline_other_kernels"] = others
        line["sort"] = ("lists persist between mc_coag calls (aero_state%valid_sort, "
                        "wrf_pmc_trans_aero.F90:2094): bin_sort only runs for cells whose "
                        "population something else changed")
        if resort:
            line["resort_every_step"] = {
                "ms_per_step": resort["elapsed"] / args.resort_steps * 1e3,
                "value": agg["r_n_coag"] / resort["elapsed"], "unit": "events/s",
                "steps": args.resort_steps,
                "bin_sort_ms_per_step": agg["r_ms_sort"] / world / args.resort_steps,
                "bin_sort_gbs": (algorithmic_bytes("bin_sort", args.n_spec, 1, 0, **per_launch)
                                 / max(agg["r_ms_sort"] / world / args.resort_steps * 1e-3, 1e-12) / 1e9)}
        if cpu_line is not None:
            both = [c for c in cpu_counts if c in (last_counts or {})]
            cpu_line["cells_compared_with_gpu"] = len(both)
            cpu_line["compared"] = (f"n_coag of step {args.warmup + args.steps - 1} (the last timed one) and "
                                    f"n_part after it, cells {both[0]}..{both[-1]} spread over the domain"
                                    if both else "nothing")
            cpu_line["gpu_matches_oracle_on_sample"] = bool(
                both and all(tuple(cpu_counts[c]) == last_counts[c] for c in both))
            line["cpu_baseline"] = cpu_line
            line["gpu_vs_cpu_events_per_sec"] = line["value"] / cpu_line["value"]
        print(json.dumps(line), flush=True)
    blk.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
