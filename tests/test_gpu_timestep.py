"""-m gpu: the loop body of partmc_timestep as ONE call (pmcx_timestep,
interface/wrf_pmc_driver.F90:169-254) against the oracle running the same loop per cell, the gas
half of the scenario update (driver:202-205), and the guards around them."""
import ctypes as C

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu
K_BROWN = 4


def _gas_scenario(n_cells, n_gas, rs):
    et = np.array([0.0, 120.0, 300.0, 900.0])
    es = np.array([1.0, 0.5, 2.0, 1.0])
    em = rs.uniform(0.0, 3e-9, (n_cells, len(et), n_gas))           # mol m^-2 s^-1
    dtm = np.array([0.0, 200.0, 1000.0])
    dr = rs.uniform(0.0, 2e-4, (n_cells, len(dtm)))                  # 1/s
    bg = rs.uniform(0.0, 40.0, (n_cells, len(dtm), n_gas))           # ppb
    return et, es, em, dtm, dr, bg


def _oracle_gas(oracle, mix, c, t, dt, T, P, h, h_old, scn):
    et, es, em, dtm, dr, bg = scn
    e = oracle.interp_1d(et, em[c], t) if et is not None else None
    s = oracle.interp_1d(et, es, t) if et is not None else 0.0
    b = oracle.interp_1d(dtm, bg[c], t) if dtm is not None else None
    r = oracle.interp_1d(dtm, dr[c], t) if dtm is not None else 0.0
    return oracle.scenario_update_gas(mix, dt, T, P, h, h_old, e, s, b, r)


def test_gas_scenario_update_parity(wp, oracle):
    """scenario_update_gas_state: times before, inside, on a node of and after the series; a
    growing layer; emission-only and dilution-only scenarios.  Bit for bit."""
    n_cells, n_gas = 37, 77                                           # gas_data.dat has 77 species
    rs = np.random.RandomState(3)
    with wp.CellBlock(n_cells, 64, 1) as blk:
        blk.set_aero_data([1800.0])
        T = rs.uniform(220, 300, n_cells); P = rs.uniform(2e4, 1.02e5, n_cells)
        blk.set_env(T, P)
        mix = rs.uniform(0, 100, (n_cells, n_gas))
        mix[:, 5] = 0.0
        blk.set_gas_state(mix)
        scn = _gas_scenario(n_cells, n_gas, rs)
        want = mix.copy()
        h = rs.uniform(20, 400, n_cells)
        for variant in ("both", "emission", "dilution"):
            if variant == "both":
                blk.set_gas_scenario(*scn); s2 = scn
            elif variant == "emission":
                blk.set_gas_scenario(scn[0], scn[1], scn[2]); s2 = (scn[0], scn[1], scn[2], None, None, None)
            else:
                blk.set_gas_scenario(None, None, None, scn[3], scn[4], scn[5])
                s2 = (None, None, None, scn[3], scn[4], scn[5])
            for t in (-5.0, 0.0, 60.0, 120.0, 250.0, 899.999, 900.0, 5000.0):
                h_old = h * rs.uniform(0.8, 1.2, n_cells)
                blk.scenario_update_gas_state(20.0, t, h, h_old)
                for c in range(n_cells):
                    want[c] = _oracle_gas(oracle, want[c], c, t, 20.0, T[c], P[c], h[c], h_old[c], s2)
                got = blk.get_gas_state()
                assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (variant, t)
        assert (want >= 0).all() and np.isfinite(want).all() and want.max() > 1.0


def test_partmc_timestep_loop(wp, oracle):
    """Seven WRF steps of 20 s with partmc_chem_dt = 60 s: mc_coag on steps 0, 3, 6 only (the
    gate, driver:189), gas and aerosol scenario updates and the rebalance on every step, in the
    reference's order -- against the oracle running that loop cell by cell."""
    n_cells, n_part, n_spec, n_class, n_gas = 5, 1500, 2, 2, 9
    seed, dt, chem_dt = 21, 20.0, 60.0
    blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, seed, K_BROWN,
                                         cap=8192, n_source=3, num_conc_scale=30.0)
    mag = info["mag"]
    # ideal counts that make the rebalance act: class 0 halves on the first step
    n_ideal = np.tile(np.array([0.3, 1.0]) * n_part / n_class, (n_cells, 1))
    blk.set_n_part_ideal(n_ideal)
    rs = np.random.RandomState(11)
    mix = rs.uniform(0, 50, (n_cells, n_gas))
    blk.set_gas_state(mix)
    scn = _gas_scenario(n_cells, n_gas, rs)
    blk.set_gas_scenario(*scn)
    em = [dict(gm_radius=2e-8, gsd_log10=0.2, vol_frac=[0.5, 0.5], wclass=0, source=1),
          dict(gm_radius=1e-7, gsd_log10=0.1, vol_frac=[0.0, 1.0], wclass=1, source=2)]
    bg = [dict(gm_radius=5e-8, gsd_log10=0.3, vol_frac=[1.0, 0.0], wclass=0, source=0)]
    em_rate = np.array([[mag[c, 0] * 1.5, mag[c, 1] * 0.3] for c in range(n_cells)])
    bg_conc = np.array([[mag[c, 0] * 700.0] for c in range(n_cells)])
    dil = np.linspace(5e-4, 3e-3, n_cells)
    height = np.linspace(30.0, 300.0, n_cells)
    want_mix = mix.copy()
    n_chem = 0
    for k in range(7):
        t = k * dt
        h_old = height * (1.0 - 0.02 * (k % 2))
        blk.partmc_timestep(t, dt, chem_dt, do_coagulation=True, kernel_type=K_BROWN,
                            random_seed=seed, height=height, old_height=h_old, emission=em,
                            emission_rate=em_rate, dilution_rate=dil, background=bg,
                            background_conc=bg_conc)
        ns, nc, st = blk.coag_counters()
        chem = (t % chem_dt) == 0.0
        n_chem += chem
        for c, oc in enumerate(cells):
            if chem:
                ost, ons, onc = oc.mc_coag(K_BROWN, chem_dt, seed, oc.cell_id, int(t / chem_dt))
                assert (ons, onc) == (int(ns[c]), int(nc[c])), (k, c)
            want_mix[c] = _oracle_gas(oracle, want_mix[c], c, t, dt, info["temp"][c], info["pres"][c],
                                      height[c], h_old[c], scn)
            oe = [dict(m, num_conc=em_rate[c, j]) for j, m in enumerate(em)]
            ob = [dict(m, num_conc=bg_conc[c, j]) for j, m in enumerate(bg)]
            _, oc.next_id, _, _, _ = oc.scenario_update(dt, t, seed, oc.cell_id, k, oe, dil[c], ob,
                                                        oc.next_id, height[c], h_old[c])   # (the layer grows on odd steps)
            _, oc.next_id = oc.rebalance(n_ideal[c], True, True, seed, oc.cell_id, k, oc.next_id)
        ok, msg = parity.compare_state(blk, cells)
        assert ok, f"WRF step {k}: {msg}"
        assert np.array_equal(blk.get_weights(), np.array([oc.mag for oc in cells])), k
        assert np.array_equal(blk.get_gas_state().view(np.uint64), want_mix.view(np.uint64)), k
        assert np.array_equal(blk.get_next_id(), [oc.next_id for oc in cells]), k
    assert n_chem == 3
    assert blk.get_weights()[0, 0] > mag[0, 0] * 1.9          # class 0 was halved
    blk.close()


def test_non_chemistry_step_does_not_coagulate(wp, oracle):
    """On a WRF step that is not a multiple of partmc_chem_dt nothing coagulates and the
    populations are untouched by anything but the rebalance (here: nothing)."""
    blk, cells, info = parity.setup_pair(wp, oracle, 3, 800, 1, 1, 4, K_BROWN, n_source=0)
    blk.set_n_part_ideal(np.full((3, 1), 800.0))
    before = blk.download_cells()
    blk.partmc_timestep(20.0, 20.0, 60.0, random_seed=4)
    after = blk.download_cells()
    for k in ("vol", "id", "least_ct"):
        assert np.array_equal(before[k], after[k])
    blk.partmc_timestep(60.0, 20.0, 60.0, random_seed=4)
    assert blk.aero_state_total_particles().sum() < 3 * 800
    blk.close()


def test_rebalance_needs_the_ideal_count(wp):
    """aero_state_rebalance with n_part_ideal never set would halve every cell away: refused."""
    with wp.CellBlock(2, 256, 1) as blk:
        blk.set_aero_data([1800.0]); blk.set_env(288.0, 1e5); blk.set_weights(1e7)
        blk.generate_lognormal(200, [dict(frac=1.0, gm_radius=2e-8, gsd_log10=0.25, vol_frac=[1.0])], 1)
        with pytest.raises(wp.PmcxError, match="pmcx_set_n_part_ideal"):
            blk.aero_state_rebalance(True, True)
        with pytest.raises(wp.PmcxError, match="pmcx_set_n_part_ideal"):
            blk.partmc_timestep(0.0, 20.0, 60.0)
        assert (blk.aero_state_total_particles() == 200).all()


def test_cell_id_base_can_be_changed_in_place(wp, oracle):
    """pmcx_set_cell_id_base re-keys an existing context: the per-cell drop-in of INTEGRATION.md 2a
    gives every cell of the loop its own stream without rebuilding the context."""
    res = []
    for gid in (0, 7, 7, 123456):
        b2, c2, _ = parity.setup_pair(wp, oracle, 1, 900, 1, 1, 9, K_BROWN, n_source=0, cell_id_base=0)
        b2.set_cell_id_base(gid)
        ns, nc, st = b2.mc_coag(K_BROWN, 600.0, 9, 0)
        ost, ons, onc = c2[0].mc_coag(K_BROWN, 600.0, 9, gid, 0)
        assert (ons, onc) == (int(ns[0]), int(nc[0]))
        res.append((int(ns[0]), int(nc[0])))
        b2.close()
    assert res[1] == res[2] and len(set(res)) >= 2


def test_transport_import_rejects_a_bad_buffer(wp):
    """The import side of the slab transport validates what another process sent: counts beyond
    a cell's capacity, records beyond the buffer and destinations outside the row are refused
    (PMCX_EINVAL) instead of being used as indices."""
    nx, nz, ny = 3, 2, 4
    n = nx * nz * ny
    with wp.CellBlock(n, 256, 1) as blk:
        blk.set_aero_data([1800.0]); blk.set_env(288.0, 1e5); blk.set_weights(1e7)
        blk.generate_lognormal(100, [dict(frac=1.0, gm_radius=2e-8, gsd_log10=0.25, vol_frac=[1.0])], 1)
        row = nx * nz

        def begin():
            blk.transport_begin(nx, nz, ny, np.full((n, 1, 4), 0.05), np.full((n, 1, nz), 0.01),
                                np.ones(n), seed=1, step=0)
        begin()
        good = blk.transport_export(1)
        nb = good.nbytes
        host = np.zeros(nb, np.uint8)
        blk._ck(blk._lib.pmcx_device_copy(blk._h, host.ctypes.data_as(C.c_void_p), good.ptr, nb, 2))
        hdr = ((row * 4 + 15) // 16) * 16

        def try_import(buf, nbytes=None):
            d = wp.host.DeviceBuffer(blk, len(buf))
            blk._ck(blk._lib.pmcx_device_copy(blk._h, d.ptr, buf.ctypes.data_as(C.c_void_p), len(buf), 1))
            blk.transport_import(0, (d.ptr.value, len(buf) if nbytes is None else nbytes))

        bad = host.copy(); bad[:4].view(np.int32)[0] = 100000          # count beyond the capacity
        with pytest.raises(wp.PmcxError, match="claims"):
            try_import(bad)
        bad = host.copy(); bad[:4].view(np.int32)[0] = -3
        with pytest.raises(wp.PmcxError, match="claims"):
            try_import(bad)
        if nb > hdr:
            with pytest.raises(wp.PmcxError, match="bytes"):
                try_import(host, nbytes=nb - 8)                         # records beyond the buffer
            bad = host.copy(); bad[hdr + 4:hdr + 8].view(np.int32)[0] = 1 << 20   # destination
            with pytest.raises(wp.PmcxError, match="outside this block"):
                try_import(bad)
        with pytest.raises(wp.PmcxError, match="header"):
            try_import(host[:8])
        try_import(host)                                                 # the untouched buffer is fine
        blk.transport_import(1, None)
        blk.transport_finish()
        # the halo scratch is reused from step to step (it used to grow by a block per call)
        m0 = blk.memory_bytes()
        for _ in range(3):
            begin(); blk.transport_import(0, None); blk.transport_import(1, None); blk.transport_finish()
        assert blk.memory_bytes() == m0


def test_transport_driver_calls_refuse_bad_input(wp):
    """The error behaviour of the transport driver's entry points: every misuse comes back as a
    PmcxError with the reason (the reference aborts; the C-ABI returns a status and a message)."""
    from tests import trans_fields as tf
    nx, nz, ny = 3, 4, 3
    n = nx * nz * ny
    with wp.CellBlock(n, 512, 1) as blk:
        blk.set_aero_data([1800.0]); blk.set_env(285.0, 9e4)
        blk.set_weights(np.full((n, 1), 1e6))
        blk.generate_lognormal(300, [dict(frac=1.0, gm_radius=5e-8, gsd_log10=0.2, vol_frac=[1.0])], seed=1)
        cv = np.ones(n)
        with pytest.raises(wp.PmcxError, match="n_part_ideal"):
            blk.transport_preweight(nx, nz, ny, 0.01, 0.01, cv)                 # ideal counts not set
        blk.set_n_part_ideal(np.full((n, 1), 300.0))
        with pytest.raises(wp.PmcxError, match="built"):
            blk.transport_preweight(nx, nz, ny, None, None, cv)                 # nothing built yet
        with pytest.raises(wp.PmcxError, match="built"):
            blk.transport(nx, nz, ny, None, None, cv)
        with pytest.raises(wp.PmcxError):
            blk.transport(nx, nz, ny, np.full((n, 1, 4), 0.01), None, cv)       # one of the two
        F = tf.make_fields(nx, nz, ny, 1, seed=0)
        with pytest.raises(ValueError, match="shape"):
            blk.transport_probs(nx, nz, ny, 20.0, 1e3, 1e3, dict(F, alt=F["alt"][1:]))
        with pytest.raises(wp.PmcxError, match="positive"):
            blk.transport_probs(nx, nz, ny, 0.0, 1e3, 1e3, F)
        with pytest.raises(wp.PmcxError, match="n_cells"):
            blk.transport_probs(nx, nz, ny + 1, 20.0, 1e3, 1e3, tf.make_fields(nx, nz, ny + 1, 1))
        blk.transport_probs(nx, nz, ny, 20.0, 1e3, 1e3, F, download=False)
        with pytest.raises(wp.PmcxError, match="built"):
            blk.transport(nx, nz + 1, ny, None, None, cv)                       # built for another grid
        blk.transport_preweight(nx, nz, ny, None, None, cv)                     # leaves them in place
        blk.transport(nx, nz, ny, None, None, cv)                               # consumes them
        with pytest.raises(wp.PmcxError, match="built"):
            blk.transport(nx, nz, ny, None, None, cv)
        with pytest.raises(wp.PmcxError, match="itself"):
            blk.boundary_copy([1], [1])
        with pytest.raises(wp.PmcxError, match="twice"):
            blk.boundary_copy([1, 2], [0, 0])
        with pytest.raises(wp.PmcxError, match="both"):
            blk.boundary_copy([1, 2], [2, 3])
        with pytest.raises(wp.PmcxError, match="outside"):
            blk.boundary_copy([1], [n])
        bg = [dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[1.0])]
        with pytest.raises(wp.PmcxError, match="twice"):
            blk.boundary_resample([4, 4], 0, bg, np.full((2, 1), 1e8), 1.1)
        with pytest.raises(wp.PmcxError, match="density"):
            blk.boundary_resample([4], 0, bg, np.full((1, 1), 1e8), 0.0)
        blk.set_transport_lateral(True)
        with pytest.raises(wp.PmcxError, match="lateral"):
            blk.transport(nx, nz, ny, 0.01, 0.01, cv, 0, 0, True, True)         # needs both non-periodic
        with pytest.raises(wp.PmcxError, match="lateral"):
            blk.set_transport_lateral(True, 2, 0)
        blk.set_transport_lateral(False)
        blk.transport(nx, nz, ny, 0.01, 0.01, cv, 0, 1, True, True)
        assert not blk.coag_counters()[2].any()


def test_asynchronous_stage_calls_give_the_same_step(wp):
    """pmcx_set_async: the stage calls of a step only enqueue (inputs and results in page-locked
    arrays, one synchronize at the end of the step); particles, weights, gases and the per-cell
    results are those of the same calls made one by one, bit for bit -- also when a later step
    overwrites the staging arrays the way a time loop does."""
    from tests import trans_fields as tf
    nx, nz, ny, n_part = 6, 5, 4, 400
    n = nx * nz * ny
    dens = np.array([1800.0, 1200.0])
    modes = [dict(frac=0.7, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0, 0.0], wclass=0, source=0),
             dict(frac=0.3, gm_radius=1.2e-7, gsd_log10=0.15, vol_frac=[0.4, 0.6], wclass=0, source=1)]
    emission = [dict(gm_radius=2e-8, gsd_log10=0.2, vol_frac=[1.0, 0.0], wclass=0, source=1)]
    rs = np.random.RandomState(3)
    surface = ((np.arange(n) // nx) % nz == 0).astype(np.int32)
    results = []
    for use_async in (False, True):
        blk = wp.CellBlock(n, 1024, 2, n_class=1, n_source=2)
        blk.set_aero_data(dens)
        blk.set_env(285.0, 9.5e4, 0.5)
        blk.set_weights(np.full((n, 1), 1e9 / n_part))
        blk.set_n_part_ideal(np.full((n, 1), float(n_part)))
        blk.set_next_id((np.arange(n, dtype=np.int64) << 32) + (1 << 20))
        blk.generate_lognormal(n_part, modes, seed=5)
        mk = (lambda a: a)
        if use_async:
            blk.reuse_outputs = True
            blk.set_async(True)

            def mk(a):
                out = blk.host_array(np.shape(a), np.asarray(a).dtype)
                out[...] = a
                return out
        cv, ra, ust, alpha, big_a, height, rrho, rate = (mk(np.zeros(n)) for _ in range(8))
        rate = mk(np.zeros((n, 1)))
        F0 = tf.make_fields(nx, nz, ny, 1, seed=1, k_scale=6.0, flux_scale=0.004 * 1e9)
        F = {k: mk(np.ascontiguousarray(v, dtype=np.float32)) for k, v in F0.items()}
        per_step = []
        for step in range(4):
            r = np.random.RandomState(100 + step)          # new inputs every step, in the same arrays
            Fs = tf.make_fields(nx, nz, ny, 1, seed=10 + step, k_scale=6.0, flux_scale=0.004 * 1e9)
            for k in F:
                F[k][...] = Fs[k]
            cv[:] = r.uniform(0.9e9, 1.1e9, n); ra[:] = r.uniform(20, 60, n); ust[:] = r.uniform(0.2, 0.5, n)
            alpha[:] = 1.2; big_a[:] = 0.002; height[:] = 50.0; rrho[:] = r.uniform(0.8, 0.9, n)
            rate[:, 0] = np.where(surface == 1, r.uniform(1e5, 3e5, n), 0.0)
            blk.transport_probs(nx, nz, ny, 20.0, 4000.0, 4000.0, F, download=False)
            ex, nrm = blk.transport_preweight(nx, nz, ny, None, None, cv, seed=2, step=step)
            nm, na = blk.transport(nx, nz, ny, None, None, cv, seed=2, step=step)
            nd, bad = blk.dry_deposition(20.0, ra, ust, alpha, big_a, height, rrho, seed=2, step=step, surface=surface)
            ne, ni, no = blk.scenario_update_aero_state(20.0, 20.0 * step, seed=2, step=step, emission=emission,
                                                        emission_rate=rate)
            if step % 2 == 0:
                blk.mc_coag(wp.COAG_KERNEL_TYPE_BROWN, 40.0, 2, step // 2, counters=False)
            blk.aero_state_rebalance(True, True, 2, step)
            blk.synchronize()
            per_step.append([np.array(x) for x in (ex, nrm, nm, na, nd, bad, ne)])
        results.append((blk.download_cells(), blk.get_weights(), per_step))
        blk.close()
    (a, wa, pa), (b, wb, pb) = results
    for k in ("part_offset", "vol", "wclass", "id", "least_ct", "greatest_ct", "n_orig"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(wa, wb)
    for step, (x, y) in enumerate(zip(pa, pb)):
        for q, (u, v) in enumerate(zip(x, y)):
            assert np.array_equal(u, v), (step, q)
    assert sum(int(x[2].sum()) for x in pa) > 1000 and sum(int(x[4].sum()) for x in pa) > 0


def test_a_step_keeps_what_every_stage_reported(wp, oracle):
    """pmcx_timestep runs mc_coag, the scenario updates and the rebalance on ONE status word: the
    event-log overflow of mc_coag is still there after the stages behind it (each of them, called
    by itself, starts a new word), and a step without coagulation does not inherit it."""
    blk, cells, info = parity.setup_pair(wp, oracle, 4, 1500, 1, 1, 6, K_BROWN, n_source=1, num_conc_scale=300.0,
                                         max_events=8, cap=2048)
    blk.set_n_part_ideal(np.full((4, 1), 1500.0))
    emission = [dict(gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0], wclass=0, source=0)]
    blk.partmc_timestep(0.0, 20.0, 60.0, random_seed=6, emission=emission, emission_rate=np.full((4, 1), 1e4))
    ns, nc, st = blk.coag_counters()
    assert (st & 1).all() and (nc == 8).all()            # every cell filled its event log of 8
    blk.partmc_timestep(20.0, 20.0, 60.0, random_seed=6, emission=emission, emission_rate=np.full((4, 1), 1e4))
    assert not blk.coag_counters()[2].any()              # (not a chemistry step: a clean word)
    blk.close()
