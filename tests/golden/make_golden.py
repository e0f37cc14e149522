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
    print("golden fixtures written")


if __name__ == "__main__":
    main()
