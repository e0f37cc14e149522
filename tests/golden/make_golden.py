"""Generates the committed fixtures in tests/golden/ from the CPU oracle.

The reference ships no golden vectors for this path and its
PartMC submodule is absent, so these are regression pins of the oracle (and,
through the parity tests, GPU targets), not outputs of the reference.
Run: python -m tests.golden.make_golden
"""
import hashlib
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def run_config1(oracle):
    """BASELINE config 1 shape: 1 cell, 1000 particles, Brownian, idealised IC
    (interface/wrf_pmc_init.F90:1766-1770: log-normal, radius 2e-8 m,
    log10 sigma 0.25, species 1), N = 1e10 m^-3, SO4 density
    (aero_data.dat:2), 288.15 K, 101325 Pa, dt 60 s, seed 0."""
    from wrf_partmc_amd.data import idealised_modes
    n_part = 1000
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    dens = np.array([1770.0])
    c = oracle.Cell(1, 1, 1, 1024, dens, ve, 288.15, 101325.0)
    c.mag[:] = 1e10 / n_part
    c.generate(idealised_modes(1), n_part, 0, 0)
    c.kmax[:] = oracle.kmax_table(4, vs, 1770.0, 1770.0, 288.15, 101325.0)
    counters = []
    for step in range(10):
        st, ns, nc = c.mc_coag(4, 60.0, 0, 0, step)
        counters.append([st, ns, nc])
    n = int(c.n_part[0])
    return dict(counters=counters, n_part=n,
                vol_checksum=hashlib.sha256(np.ascontiguousarray(c.vol[:, :n]).tobytes()).hexdigest(),
                id_checksum=hashlib.sha256(np.ascontiguousarray(c.id[:n]).tobytes()).hexdigest())


def run_grid(oracle, sha=True):
    """A 3 x 2 x 3 doubly periodic grid, 300 particles per cell, two weight classes, two species,
    two sources, component lists of up to 4 entries: three steps of preweight -> transport ->
    mc_coag (Brownian, 1800 s) -> rebalance, seed 5.  Pins the transport's draw (8-bit prefixes,
    round 3), the exact-count thinning, the order of arrivals, the ids of copies and the merge
    order of the component lists in one set of checksums.  Returns counters and checksums (and,
    sha=False, the cells)."""
    from tests import parity
    nx, nz, ny, n_class, n_spec, n_part, cap = 3, 2, 3, 2, 2, 300, 1024
    n = nx * nz * ny
    names, dens = parity.default_species(n_spec)
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    modes, num_conc = parity.make_modes(n_spec, n_class)
    for m in modes:
        m["source"] = m["source"] % 2
    rng = np.random.RandomState(5)
    mag0 = num_conc * 40.0 / n_part * (1.0 + 0.5 * np.arange(n_class))
    cells = []
    for c in range(n):
        oc = oracle.Cell(n_spec, n_class, 2, cap, dens, ve, 285.0, 9.5e4, max_events=1024, max_appends=256,
                         n_comp_max=4)
        oc.mag[:] = mag0 * rng.uniform(0.8, 1.25)
        oc.generate(modes, n_part, 5, c, first_id=c << 32, create_time=float(c))
        oc.lct[:n_part] = oc.gct[:n_part] = oc.comp_time[:n_part, 0] = c + 1e-3 * np.arange(n_part)
        oc.kmax[:] = oracle.kmax_table(4, vs, dens.min(), dens.max(), 285.0, 9.5e4)
        oc.cell_id = c
        cells.append(oc)
    cv = rng.uniform(0.8e9, 1.3e9, n)
    ph = rng.uniform(0, 0.08, (n, n_class, 4)); pv = rng.uniform(0, 0.04, (n, n_class, nz))
    n_ideal = np.full((n, n_class), 0.45 * n_part / n_class)
    next_id = (np.arange(n, dtype=np.int64) << 32) + (1 << 20)
    counters = []
    for step in range(3):
        ex, nr = oracle.transport_preweight(cells, nx, nz, ny, ph, pv, cv, n_ideal, 5, 0, step, True, True)
        st, nm, na = oracle.transport(cells, nx, nz, ny, ph, pv, cv, 5, 0, step, next_id, True, True)
        ev = []
        for c, oc in enumerate(cells):
            s2, ns, nc = oc.mc_coag(4, 1800.0, 5, c, step)
            _, next_id[c] = oc.rebalance(n_ideal[c], True, True, 5, c, step, int(next_id[c]))
            ev.append([s2, ns, nc])
        counters.append(dict(thinned=int(nr.sum()), moved=int(nm.sum()), arrived=int(na.sum()), status=int(st),
                             coag=[int(sum(e[k] for e in ev)) for k in range(3)]))
    if not sha:
        return cells, counters
    h = lambda parts: hashlib.sha256(b"".join(np.ascontiguousarray(p).tobytes() for p in parts)).hexdigest()
    live = [(oc, int(oc.n_part[0])) for oc in cells]
    return dict(counters=counters, n_part=[k for _, k in live],
                vol=h(oc.vol[:, :k] for oc, k in live), id=h(oc.id[:k] for oc, k in live),
                wclass=h(oc.wclass[:k] for oc, k in live), mag=h(oc.mag for oc, _ in live),
                comp_source=h(oc.comp_src[:k] for oc, k in live), comp_time=h(oc.comp_time[:k] for oc, k in live))


def main():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import pmco_py as oracle
    d = np.logspace(-9, -5, 16)
    tabs = {}
    for key, (T, p) in (("k_288", (288.15, 101325.0)), ("k_220", (220.0, 25000.0))):
        tabs[key] = np.array([[oracle.kernel(4, math.pi / 6 * a ** 3, math.pi / 6 * a ** 3 * 1770.0,
                                             math.pi / 6 * b ** 3, math.pi / 6 * b ** 3 * 1000.0, T, p)
                               for b in d] for a in d])
    np.savez(os.path.join(HERE, "brownian_kernel_16x16.npz"), diam=d, **tabs)
    with open(os.path.join(HERE, "single_cell_1000_seed0.json"), "w") as f:
        json.dump(run_config1(oracle), f, indent=1)
    with open(os.path.join(HERE, "grid_3x2x3_seed5.json"), "w") as f:
        json.dump(run_grid(oracle), f, indent=1)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
