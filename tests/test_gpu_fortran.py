"""-m gpu: the Fortran ISO_C_BINDING shim (wrf-partmc_amd/fortran/pmcx_shim.F90),
driven the way partmc_timestep drives mc_coag, against the CPU oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "wrf-partmc_amd", "fortran")
EXE = os.path.join(FDIR, "test_shim")


def _build():
    if not os.path.exists(EXE) and shutil.which("make") and os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        subprocess.call(["make", "-C", FDIR, "-s"])
    if not os.path.exists(EXE):
        pytest.skip("flang or the prebuilt test_shim is not available")


@pytest.mark.parametrize("mode", [1, 0])
def test_fortran_shim_matches_oracle(tmp_path, wp, oracle, mode):
    _build()
    n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps = 3, 900, 2, 2, 3, 4, 2
    seed, del_t = 12345, 900.0
    names, dens = parity.default_species(n_spec)
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    # the product's tables are what both sides must use
    with wp.CellBlock(1, 64, n_spec) as blk:
        ve, vs = blk.bin_tables()
        const_k, beta = blk.cfg.const_k, blk.cfg.additive_beta
        max_events, max_appends = blk.cfg.max_events, blk.cfg.max_appends
    modes, num_conc = parity.make_modes(n_spec, n_class)
    cells = []
    buf = struct.pack("<7i", n_cells, n_part, n_spec, n_class, n_source, kernel, n_steps)
    buf += struct.pack("<q", seed) + struct.pack("<d", del_t) + dens.astype("<f8").tobytes()
    for c in range(n_cells):
        T, p = 288.15 - 5 * c, 101325.0 - 8000 * c
        oc = oracle.Cell(n_spec, n_class, n_source, 1024, dens, ve, T, p,
                         max_events=max_events, max_appends=max_appends)
        oc.mag[:] = num_conc / n_part * (1 + 0.5 * np.arange(n_class))
        oc.generate(modes, n_part, seed, c, first_id=c << 32, create_time=float(c))
        oc.kmax[:] = oracle.kmax_table(kernel, vs, dens.min(), dens.max(), T, p, const_k, beta)
        cells.append(oc)
        buf += struct.pack("<2d", T, p) + oc.mag.astype("<f8").tobytes()
        for i in range(n_part):
            buf += oc.vol[:, i].astype("<f8").tobytes() + oc.n_orig[:, i].astype("<i4").tobytes()
            buf += struct.pack("<i", int(oc.wclass[i]) + 1) + struct.pack("<q", int(oc.id[i]))
            buf += struct.pack("<2d", oc.lct[i], oc.gct[i])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(buf)
    r = subprocess.run([EXE, str(fin), str(fout), str(mode)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = fout.read_bytes()
    off = 0
    for step in range(n_steps):
        ns = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        nc = np.frombuffer(out, "<i4", n_cells, off); off += 4 * n_cells
        for c, oc in enumerate(cells):
            # the shim uploads the Fortran aero_state before every call: caches are
            # recomputed from the species and the sort is not valid (a host-side edit)
            oc.refresh_caches()
            st, ons, onc = oc.mc_coag(kernel, del_t, seed, c, step, const_k, beta)
            assert (int(ns[c]), int(nc[c])) == (ons, onc), (step, c)
    rec = 8 * n_spec + 4 * n_source + 4 + 8 + 16
    for c, oc in enumerate(cells):
        n = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
        assert n == int(oc.n_part[0])
        blob = np.frombuffer(out, np.uint8, rec * n, off).reshape(n, rec); off += rec * n
        vol = blob[:, :8 * n_spec].copy().view("<f8")
        ids = blob[:, 8 * n_spec + 4 * n_source + 4:8 * n_spec + 4 * n_source + 12].copy().view("<i8")[:, 0]
        cls = blob[:, 8 * n_spec + 4 * n_source:8 * n_spec + 4 * n_source + 4].copy().view("<i4")[:, 0]
        assert np.array_equal(vol.T, oc.vol[:, :n])
        assert np.array_equal(ids, oc.id[:n])
        assert np.array_equal(cls - 1, oc.wclass[:n])
    assert off == len(out)


def test_transport_across_two_ranks():
    """The N > 1 form of the transport: two processes (one slab each, sharing this GPU, gloo
    rendezvous on 127.0.0.1) exchange their boundary rows' movers with
    CellBlock.transport_distributed; every rank checks its rows against the whole-domain
    transport bit for bit (tests/dist_transport_worker.py)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PMCX_TEST_BACKEND="gloo")
    env.pop("PMCX_SAMPLER_LANES", None)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29533",
         os.path.join(ROOT, "tests", "dist_transport_worker.py")],
        capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "distributed transport ok: 2 ranks" in out.stdout
