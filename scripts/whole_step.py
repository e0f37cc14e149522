#!/usr/bin/env python3
"""One WRF-PartMC step (minus MOSAIC) that never leaves HBM: the order of the reference's
driver -- transport + dry deposition (wrf_pmc_trans_driver, interface/wrf_pmc_trans.F90:37-200,
deposition at :184), then partmc_timestep (interface/wrf_pmc_driver.F90:169-254: mc_coag at the
chemistry step, scenario_update_aero_state for emission and dilution, aero_state_rebalance) and
the partmc_process observables at output time -- on one block of cells, with the per-kernel
device time of each piece.  Synthetic inputs (probabilities, resistances, emission rates);
the point is that every piece runs on the resident state and what each costs.

  python scripts/whole_step.py [--nx 64 --ny 64 --levels 30 --particles 10000 --steps 6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrf_partmc_amd as wp  # noqa: E402
from wrf_partmc_amd.data import (aerodynamic_resistance, idealised_modes,  # noqa: E402
                                 zhang_surface_parameters)

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=64)
ap.add_argument("--ny", type=int, default=64)
ap.add_argument("--levels", type=int, default=30)
ap.add_argument("--particles", type=int, default=10000)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--dt", type=float, default=20.0)          # WRF step, s
ap.add_argument("--chem-dt", type=float, default=60.0)     # registry.partmc:24
ap.add_argument("--given-probs", action="store_true",
                help="hand the transport fixed probabilities instead of building them from WRF-shaped "
                     "fields on the device every step (transport_probs + transport_preweight)")
ap.add_argument("--trace", type=int, default=0, help="every TRACE steps: cells thinned by the preweight, weight and population spread (stderr)")
ap.add_argument("--law", type=int, default=1, help="pmcx_set_transport_law: 0 movers with geometric copies, 1 movers and emitters")
ap.add_argument("--prep", action="store_true", help="the adds' weight preparation on (pmcx_set_weight_preparation), as the Fortran drivers run")
ap.add_argument("--sync-stages", action="store_true", help="every stage call waits for the device (the default before round 3)")
ap.add_argument("--cap-factor", type=float, default=2.2, help="slots per cell / ideal count (the Fortran shim's default headroom is 3)")
ap.add_argument("--n-class", type=int, default=1, help="weight classes (class c stands for 1 + c / 2 times class 0's number, equal shares)")
ap.add_argument("--n-spec", type=int, default=1, help="species (the first carries the volume)")
ap.add_argument("--auto-grow", action="store_true", help="pmcx_set_auto_grow(0.75, 1.5): the block grows like PartMC's particle arrays")
a = ap.parse_args()

nx, ny, nz, N = a.nx, a.ny, a.levels, a.particles
n = nx * ny * nz
# slots: the rebalance lets a cell grow to twice its ideal count before it halves it
# (aero_state_rebalance), and a step's arrivals come on top
cap = (int(N * a.cap_factor) + 63) // 64 * 64
C, S = a.n_class, a.n_spec
blk = wp.CellBlock(n, cap, S, n_class=C, n_source=2, max_events=1024, max_appends=128 if C == 1 else 1024)
blk.set_aero_data([1770.0] + [1000.0 + 40.0 * s_ for s_ in range(1, S)])
# transport's cell order: c = (j * nz + k) * nx + i
k = (np.arange(n) // nx) % nz
z = np.linspace(50.0, 12000.0, nz)
temp = 288.15 - 6.5e-3 * z
pres = 101325.0 * (temp / 288.15) ** 5.2559
blk.set_env(temp[k], pres[k], np.full(n, 0.5))
w_cls = 1.0 + 0.5 * np.arange(C)
blk.set_weights(np.tile(1e10 / (N / C * w_cls.sum()) * w_cls, (n, 1)))
blk.set_n_part_ideal(np.full((n, C), float(N) / C))
if a.prep:
    blk.set_weight_preparation(True, True, True)
blk.set_transport_law(a.law)
if a.auto_grow:
    blk.set_auto_grow(True, 0.75, 1.5)
blk.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 24))
blk.generate_lognormal(N, [dict(idealised_modes(S, 1)[0], frac=1.0 / C, wclass=c_, source=0) for c_ in range(C)], seed=0)
blk.synchronize()      # (no pmcx_update_kernel_bounds: mc_coag keeps each cell's bounds current)

# per-step inputs and results live in page-locked host memory (pmcx_host_alloc), as the Fortran
# shim's staging arrays do: an upload is then one DMA, not a page-by-page staging the caller waits for
blk.reuse_outputs = True
if not a.sync_stages:
    blk.set_async(True)      # a step is one uninterrupted queue; results are read after its synchronize()


def pinned(a):
    a = np.asarray(a)
    out = blk.host_array(a.shape, a.dtype)
    out[...] = a
    return out


rng = np.random.RandomState(1)
prob_h = np.full((n, C, 4), 0.01) * rng.uniform(0.9, 1.1, (n, C, 4))     # Courant-like numbers
prob_v = np.zeros((n, C, nz))
for kk in range(nz):                                                      # to the level above/below
    m = k == kk
    if kk + 1 < nz:
        prob_v[m, :, kk + 1] = 0.004
    if kk > 0:
        prob_v[m, :, kk - 1] = 0.004
cell_volume = np.full(n, 1.0e9)
surface = (k == 0).astype(np.int32)
ustar = np.full(n, 0.35)
ra = np.broadcast_to(aerodynamic_resistance(0.0, 0.1, 0.35), (n,))
alpha, big_a = zhang_surface_parameters(np.full(n, 7))                    # grassland
height = np.full(n, 50.0)
rrho = 1.0 / (pres[k] / (287.0 * temp[k]))
emission = [dict(num_conc=None, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0] + [0.0] * (S - 1), wclass=0, source=1)]
emit_rate = np.where(surface[:, None] == 1, 2.0e5, 0.0)                   # m^-3 s^-1 at the ground

fields = None
if not a.given_probs:
    # WRF-shaped single-precision fields (tests/trans_fields.py); fluxes scaled to the cells'
    # mixing ratios so that a few particles in a thousand leave through a face per step
    from tests import trans_fields
    fields = trans_fields.make_fields(nx, nz, ny, C, seed=3, k_scale=6.0, flux_scale=0.004 * 1e10)
    for key in ("u_flux", "v_flux", "w_flux"):     # (a class's flux in proportion to the number it stands for)
        fields[key] = fields[key] * (w_cls / w_cls.sum()).reshape((C,) + (1,) * (fields[key].ndim - 1))

cell_volume, surface, ustar, ra, alpha, big_a, height, rrho, emit_rate = (
    pinned(np.ascontiguousarray(x)) for x in (cell_volume, surface, ustar, ra, alpha, big_a, height, rrho, emit_rate))
env_t, env_p, env_rh = pinned(temp[k]), pinned(pres[k]), pinned(np.full(n, 0.5))
if fields is not None:
    for key in ("alt", "xkhh", "u_flux", "v_flux", "w_flux", "z", "z_at_w", "exch_h"):
        fields[key] = pinned(np.ascontiguousarray(fields[key], dtype=np.float32))

blk.set_aero_scenario_modes(0, emission)   # the emission's modes stay on the device: only the rates travel

names = ("transport_probs", "transport_preweight", "transport", "dry_deposition", "scenario", "kernel_bounds", "bin_sort", "coag_sample", "compact",
         "rebalance", "diagnostics")
blk.profile_enable(True)
blk.profile_reset()
t0 = time.perf_counter()
elapsed = 0.0
moved = deposited = emitted = events = 0
host = {"transport_probs": 0.0, "transport_preweight": 0.0, "transport": 0.0, "dry_deposition": 0.0, "scenario": 0.0, "mc_coag": 0.0, "rebalance": 0.0, "checks": 0.0}
step_wall = []
for step in range(a.steps):
    ts = time.perf_counter()
    if fields is not None:
        # wrf_pmc_trans_driver: probabilities from the fields and the populations, preweight, sampling
        blk.transport_probs(nx, nz, ny, a.dt, 4000.0, 4000.0, fields, download=False)
        host["transport_probs"] += time.perf_counter() - ts
        tp = time.perf_counter()
        ex_pw, nr_pw = blk.transport_preweight(nx, nz, ny, None, None, cell_volume, seed=0, step=step)
        if a.trace and step % a.trace == 0:
            blk.synchronize()
            w_, np_ = blk.get_weights()[:, 0], blk.aero_state_total_particles()
            print(f"step {step}: preweight thinned {int((nr_pw > 0).sum())} cells ({int(nr_pw.sum())} particles); weights "
                  f"{w_.min():.3g} .. {w_.max():.3g} (median {np.median(w_):.3g}); particles per cell {np_.min()} .. {np_.max()} "
                  f"(median {int(np.median(np_))})", file=sys.stderr)
        host["transport_preweight"] += time.perf_counter() - tp
        tp = time.perf_counter()
        nm, na = blk.transport(nx, nz, ny, None, None, cell_volume, seed=0, step=step)
        host["transport"] += time.perf_counter() - tp
    else:
        nm, na = blk.transport(nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=step)
        host["transport"] += time.perf_counter() - ts
    if a.trace and step % a.trace == 0:
        blk.synchronize()
        print(f"step {step}: {int(nm.sum())} movers, {int(na.sum())} arrivals ({na.sum() / max(nm.sum(), 1):.2f} per mover), "
              f"most into one cell {int(na.max())}", file=sys.stderr)
    tc = time.perf_counter()
    if step == 0:
        blk.synchronize()
        st_or = blk.coag_totals()[3]                   # (one word: the OR of every cell's status bits)
        assert not st_or, f"step {step}: status bits {st_or} after the transport (16: a cell ran out of slots)"
        if (blk.get_weights() == 1e10 / N).all():      # equal weights: every mover arrives as one copy
            assert nm.sum() == na.sum(), "transport lost particles"
    t1 = time.perf_counter(); host["checks"] += t1 - tc
    nr, bad = blk.dry_deposition(a.dt, ra, ustar, alpha, big_a, height, rrho, seed=0, step=step,
                                 surface=surface)
    t2 = time.perf_counter(); host["dry_deposition"] += t2 - t1
    ne, ni, no = blk.scenario_update_aero_state(a.dt, elapsed, seed=0, step=step, emission=len(emission),
                                                emission_rate=emit_rate)    # (the modes are resident)
    t3 = time.perf_counter(); host["scenario"] += t3 - t2
    if elapsed % a.chem_dt == 0.0:
        # WRF hands PartMC new air every step (wrf_to_partmc, driver:611-637): the kernel bounds
        # of every cell are computed again inside mc_coag
        drift = 0.02 * np.sin(0.1 * step + 0.001 * np.arange(n))
        env_t[:] = temp[k] + drift
        blk.set_env(env_t, env_p, env_rh)
        blk.mc_coag(wp.COAG_KERNEL_TYPE_BROWN, a.chem_dt, 0, int(elapsed // a.chem_dt), counters=False)
    t4 = time.perf_counter(); host["mc_coag"] += t4 - t3
    blk.aero_state_rebalance(True, True, 0, step)
    blk.synchronize()
    if elapsed % a.chem_dt == 0.0:
        _, nc_tot, _, st_or = blk.coag_totals()       # (sums and the OR of the status bits, reduced on the device)
        events += int(nc_tot)
        assert not (st_or & ~4), "status"
    host["rebalance"] += time.perf_counter() - t4
    step_wall.append((time.perf_counter() - ts) * 1e3)
    if step == 0:
        host0 = dict(host)
    moved += int(nm.sum()); deposited += int(nr.sum()); emitted += int(ne.sum())
    elapsed += a.dt
tn, tm, nd, md = blk.diagnostics()
blk.synchronize()
wall = time.perf_counter() - t0
out = {"cells": n, "particles_per_cell": N, "steps": a.steps, "wall_ms_per_step": wall / a.steps * 1e3,
       "moved_per_step": moved / a.steps, "deposited_per_step": deposited / a.steps,
       "emitted_per_step": emitted / a.steps, "coag_events": events,
       "particles_end": int(blk.aero_state_total_particles().sum()),
       "wall_ms_each_step": [round(x, 2) for x in step_wall],
       "host_wall_ms_per_step": {k: v / a.steps * 1e3 for k, v in host.items()},
       "host_wall_ms_steady": {k: (v - host0[k]) / max(a.steps - 1, 1) * 1e3 for k, v in host.items()},
       "device_ms": {}}
for name in names:
    ms, cnt = blk.profile_get(name)
    if cnt:
        out["device_ms"][name] = {"per_launch": ms / cnt, "launches": int(cnt)}
print(json.dumps(out))
blk.close()
