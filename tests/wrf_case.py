"""A small WRF-PartMC case for the two drop-in drivers of the Fortran shim
(partmc_timestep_pmcx, wrf_pmc_trans_driver_pmcx): the input files wrf-partmc_amd/fortran/
test_driver.F90 reads, and the CPU oracle running the same loop -- interface/wrf_pmc_driver.F90:
46-290 followed by interface/wrf_pmc_trans.F90:37-200 per WRF step -- cell by cell.

Test infrastructure (imports the oracle)."""
import math
import struct

import numpy as np

K_BROWN = 4
F32 = np.float32


class Case:
    """nx x nz x ny cells in WRF's order c = (j nz + k) nx + i, open lateral boundaries unless
    periodic; emissions in the lowest level, a background (dilution + the inflow boundary
    condition), gases, sea salt over the ocean columns, transport from WRF-shaped fields, dry
    deposition on the lowest level."""

    def __init__(self, nx=4, nz=3, ny=4, n_part=300, n_steps=7, periodic=False, seed=91, sea_salt_opt=1, n_class=2,
                 n_spec=3):
        self.nx, self.nz, self.ny, self.n_part, self.n_steps = nx, nz, ny, n_part, n_steps
        self.periodic, self.seed, self.sea_salt_opt = periodic, seed, sea_salt_opt
        self.n_cells = nx * nz * ny
        self.n_spec, self.n_class, self.n_source, self.n_gas = n_spec, n_class, 3, 5
        self.i_na, self.i_cl = 2, 3                         # 1-based species of the sea-salt source
        self.dt, self.chem_dt, self.dx, self.dy = 20.0, 60.0, 3000.0, 3000.0
        self.vertmix = 1
        self.num_particles = float(n_part)
        # (more than three species: the densities of WRFV3/test/em_scm_xy/aero_data.dat:2-21 behind the first three)
        scm = [1770.0, 1770.0, 1900.0, 1770.0, 1800.0, 1400.0, 1400.0, 1400.0, 1400.0, 1000.0, 1400.0, 1400.0, 1400.0,
               2600.0, 1900.0, 2600.0, 2600.0, 1000.0, 1700.0, 1000.0]
        self.density = np.array(([1770.0, 1900.0, 1900.0] + scm[3:] + [1500.0] * n_spec)[:n_spec])
        r = self.r = np.random.RandomState(seed)
        n = self.n_cells
        # (more than two classes -- the registry carries 40, registry.partmc_trans:2-41: the initial particles sit
        # in the first two, emissions, background and sea salt bring others to life, most stay empty)
        ideal = np.array([0.55, 0.5] + [0.12] * (n_class - 2))[:n_class] if n_class >= 2 else np.array([1.05])
        self.n_ideal = np.tile(ideal * n_part, (n, 1))
        self.num_conc0 = 3.0e9
        self.mag = (self.num_conc0 / n_part) * np.tile(1.0 + 0.5 * np.arange(self.n_class), (n, 1))
        self.gas0 = r.uniform(0.0, 60.0, (n, self.n_gas))
        lev = (np.arange(n) // nx) % nz
        ii, jj = np.arange(n) % nx, np.arange(n) // (nx * nz)
        self.lev, self.ii, self.jj = lev, ii, jj
        self.inside = np.ones(n, bool) if periodic else ((ii > 0) & (ii < nx - 1) & (jj > 0) & (jj < ny - 1))
        # scenario: emissions where the ground is (lowest level), two time levels that differ
        self.n_time_e, self.n_mode_e, self.n_time_b, self.n_mode_b = 2, 2, 2, 2
        self.has_em = (lev == 0)
        self.em_time = np.array([0.0, 60.0])
        self.em_scale = np.array([1.0, 0.6])
        self.bg_time = np.array([0.0, 80.0])
        self.em, self.bg = [], []
        self.bg_dil = r.uniform(2e-4, 2e-3, (n, 2))
        self.gas_dil = r.uniform(0.0, 3e-4, (n, 2))
        self.gas_em = r.uniform(0.0, 3e-9, (n, 2, self.n_gas))
        self.gas_bg = r.uniform(0.0, 40.0, (n, 2, self.n_gas))
        for c in range(n):
            def modes(conc_scale, radii):
                out = []
                for t in range(2):
                    ms = []
                    for m in range(2):
                        vf = r.uniform(0.1, 1.0, self.n_spec); vf /= vf.sum()
                        ms.append(dict(num_conc=conc_scale[m] * r.uniform(0.6, 1.4),
                                       gm_radius=radii[m] * r.uniform(0.8, 1.25), gsd_log10=r.uniform(0.12, 0.3),
                                       vol_frac=vf, wclass=(m % self.n_class if self.n_class <= 2 else (4 * m + 1) % self.n_class),
                                       source=(m + t) % self.n_source))
                    out.append(ms)
                return out
            # emission: number flux per m^2 and s such that a few particles per step appear
            self.em.append(modes([60.0 * self.mag[c, 0] * 40.0 / 20.0, 20.0 * self.mag[c, 1] * 40.0 / 20.0],
                                 [3e-8, 1.2e-7]))
            # background: number concentrations near the populations' (the resample's target)
            self.bg.append(modes([0.5 * n_part * self.mag[c, 0], 0.45 * n_part * self.mag[c, 1]], [4e-8, 1.5e-7]))
        self._fields()

    # ---- WRF's fields, one set per step -------------------------------------------------
    def _fields(self):
        from tests import trans_fields as tf
        nx, nz, ny, Cn = self.nx, self.nz, self.ny, self.n_class
        r = self.r
        self.F, self.G = [], []
        flux_scale = 0.05 * self.n_part * self.mag[0, 0]
        ocean = r.uniform(size=(ny + 2, nx + 2)) < 0.6
        triples = [(3.0, 4.0), (6.0, 8.0), (5.0, 12.0), (8.0, 15.0)]    # wind speeds exact in any arithmetic
        pick = r.randint(0, 4, (ny + 2, nx + 2))
        u_prev = v_prev = None
        for s in range(self.n_steps):
            F = tf.make_fields(nx, nz, ny, Cn, seed=100 + s, k_scale=6.0, flux_scale=flux_scale)
            if s:        # the terrain does not move
                for k in ("z", "z_at_w"):
                    F[k] = self.F[0][k]
            g = dict(time=float(s) * self.dt)
            sh3 = (ny + 2, nz + 1, nx + 2)
            for name in ("alt", "xkhh"):
                a = np.full(sh3, 0.85, F32); a[:, :nz, :] = F[name]; g[name] = a
            for name in ("z", "z_at_w", "exch_h"):
                a = np.zeros(sh3, F32)
                kk = F[name].shape[1]
                a[1:ny + 1, :kk, 1:nx + 1] = F[name]
                g[name] = a
            zc = 0.5 * (g["z_at_w"][:, 1:, :] + g["z_at_w"][:, :-1, :])
            pb = np.full(sh3, 1.0e5, F32); pb[:, :nz, :] = (1.0e5 * np.exp(-zc / 8000.0)).astype(F32)
            g["pb"] = pb
            g["p"] = (r.uniform(-300.0, 300.0, sh3)).astype(F32)
            g["t_2"] = (r.uniform(-12.0, -4.0, sh3) + 0.004 * g["z"]).astype(F32)
            g["qv"] = r.uniform(2e-3, 9e-3, sh3).astype(F32)
            shu = (ny + 3, nz + 1, nx + 3)
            u2 = r.normal(0.0, 2.0, shu).astype(F32); v2 = r.normal(0.0, 2.0, shu).astype(F32)
            u1 = u2.copy() if u_prev is None else u_prev.copy()
            v1 = v2.copy() if v_prev is None else v_prev.copy()
            g.update(u_2=u2, v_2=v2, u_1=u1, v_1=v1)
            u_prev, v_prev = u2, v2
            for name, key in (("u_flux", "u_flux"), ("v_flux", "v_flux"), ("w_flux", "w_flux")):
                a = np.zeros((Cn + 1, ny + 3, nz + 1, nx + 3), F32)
                src = F[key]
                a[1:, 1:1 + src.shape[1], :src.shape[2], 1:1 + src.shape[3]] = src
                g[name] = a
            sh2 = (ny + 2, nx + 2)
            g["xland"] = np.where(ocean, 2.0, 1.0).astype(F32)
            g["u10"] = np.array([[triples[q][0] for q in row] for row in pick], F32)
            g["v10"] = np.array([[triples[q][1] for q in row] for row in pick], F32)
            g["msftx"] = np.ones(sh2, F32); g["msfty"] = np.ones(sh2, F32)
            g["ust"] = r.uniform(0.15, 0.5, sh2).astype(F32)
            g["rmol"] = r.choice([0.0, 0.01, -0.02, 2e-7], size=sh2).astype(F32)
            g["znt"] = r.uniform(0.02, 0.5, sh2).astype(F32)
            g["ivgtyp"] = r.choice([7, 16, 1, 11, 24], size=sh2).astype(np.int32)
            self.F.append(F); self.G.append(g)
        self.ocean = ocean

    def fields_bytes(self):
        out = b""
        for g in self.G:
            out += struct.pack("<d", g["time"])
            for name in ("p", "pb", "t_2", "qv", "alt", "z", "z_at_w", "xkhh", "exch_h", "u_2", "v_2", "u_1",
                         "v_1", "u_flux", "v_flux", "w_flux", "xland", "u10", "v10", "msftx", "msfty", "ust",
                         "rmol", "znt", "ivgtyp"):
                out += np.ascontiguousarray(g[name]).tobytes()
        return out

    # ---- the initial populations (oracle cells) and the state file ------------------------
    def make_cells(self, oracle, v_edge, cap, max_events, max_appends):
        from tests import parity
        modes, _ = parity.make_modes(self.n_spec, self.n_class)
        for m in modes:
            m["source"] = m["source"] % self.n_source
        cells = []
        for c in range(self.n_cells):
            oc = oracle.Cell(self.n_spec, self.n_class, self.n_source, cap, self.density, v_edge,
                             max_events=max_events, max_appends=max_appends, n_comp_max=8)   # (the shim's default)
            oc.mag[:] = self.mag[c]
            oc.generate(modes, self.n_part, self.seed, c, first_id=((c + 1) << 40) + 7, create_time=-float(c))
            # every particle its own create time: the lists show whose constituents they hold
            oc.lct[:self.n_part] = oc.gct[:self.n_part] = oc.comp_time[:self.n_part, 0] = \
                -float(c) - 1e-3 * np.arange(self.n_part)
            oc.cell_id = c
            oc.next_id = 1 + (c << 32)          # the shim's id blocks (pmcx_upload_states)
            cells.append(oc)
        return cells

    def state_bytes(self, cells):
        buf = struct.pack("<18i", self.nx, self.nz, self.ny, self.n_part, self.n_spec, self.n_class,
                          self.n_source, self.n_gas, self.n_steps, int(self.periodic), self.sea_salt_opt,
                          self.i_na, self.i_cl, self.n_time_e, self.n_mode_e, self.n_time_b, self.n_mode_b,
                          self.vertmix)
        buf += struct.pack("<q", self.seed)
        buf += struct.pack("<5d", self.dt, self.chem_dt, self.dx, self.dy, self.num_particles)
        buf += self.density.astype("<f8").tobytes()

        def dist(ms):
            b = b""
            for m in ms:
                b += struct.pack("<3d", m["num_conc"], m["gm_radius"], m["gsd_log10"])
                b += np.asarray(m["vol_frac"], "<f8").tobytes()
                b += struct.pack("<2i", m["source"] + 1, m["wclass"] + 1)
            return b
        for c, oc in enumerate(cells):
            buf += self.mag[c].astype("<f8").tobytes() + self.n_ideal[c].astype("<f8").tobytes()
            for i in range(self.n_part):
                buf += oc.vol[:, i].astype("<f8").tobytes() + oc.n_orig[:, i].astype("<i4").tobytes()
                buf += struct.pack("<i", int(oc.wclass[i]) + 1) + struct.pack("<q", int(oc.id[i]))
                buf += struct.pack("<2d", oc.lct[i], oc.gct[i])
            buf += self.gas0[c].astype("<f8").tobytes()
            buf += struct.pack("<i", int(self.has_em[c]))
            if self.has_em[c]:
                buf += self.em_time.astype("<f8").tobytes() + self.em_scale.astype("<f8").tobytes()
                for t in range(self.n_time_e):
                    buf += dist(self.em[c][t]) + self.gas_em[c, t].astype("<f8").tobytes()
            buf += self.bg_time.astype("<f8").tobytes() + self.bg_dil[c].astype("<f8").tobytes()
            buf += self.gas_dil[c].astype("<f8").tobytes()
            for t in range(self.n_time_b):
                buf += dist(self.bg[c][t]) + self.gas_bg[c, t].astype("<f8").tobytes()
        return buf

    # ---- the driver's output ------------------------------------------------------------
    def parse_output(self, out, n_cells=None):
        """n_cells: the cells the dump holds (a rank of a cut domain writes its patch's only)."""
        n, off = (self.n_cells if n_cells is None else n_cells), 0
        steps = []
        for s in range(self.n_steps):
            env = np.frombuffer(out, "<f8", 7 * n, off).reshape(7, n); off += 8 * 7 * n
            npart = np.frombuffer(out, "<i4", n, off); off += 4 * n
            cnt = np.frombuffer(out, "<i8", 4, off); off += 32
            steps.append(dict(temp=env[0], pres=env[1], rh=env[2], height=env[3], cell_volume=env[4],
                              inv_density=env[5], aer_res=env[6], n_parts=npart, n_samp=int(cnt[0]),
                              n_coag=int(cnt[1]), n_moved=int(cnt[2]), n_deposited=int(cnt[3])))
        rec = 8 * self.n_spec + 4 * self.n_source + 4 + 8 + 16
        final = []
        for c in range(n):
            mag = np.frombuffer(out, "<f8", self.n_class, off); off += 8 * self.n_class
            gas = np.frombuffer(out, "<f8", self.n_gas, off); off += 8 * self.n_gas
            npt = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
            blob = np.frombuffer(out, np.uint8, rec * npt, off).reshape(npt, rec); off += rec * npt
            o = 8 * self.n_spec
            final.append(dict(mag=mag, gas=gas, n=npt, vol=blob[:, :o].copy().view("<f8").T,
                              n_orig=blob[:, o:o + 4 * self.n_source].copy().view("<i4").T,
                              wclass=blob[:, o + 4 * self.n_source:o + 4 * self.n_source + 4].copy().view("<i4")[:, 0] - 1,
                              id=blob[:, o + 4 * self.n_source + 4:o + 4 * self.n_source + 12].copy().view("<i8")[:, 0],
                              ct=blob[:, o + 4 * self.n_source + 12:].copy().view("<f8")))
            comp = []           # component(:) of every particle: arrays of (source_id, create_time)
            for _ in range(npt):
                k = int(np.frombuffer(out, "<i4", 1, off)[0]); off += 4
                comp.append(np.frombuffer(out, np.dtype([("s", "<i4"), ("t", "<f8")]), k, off)); off += 12 * k
            final[-1]["comp"] = comp
        assert off == len(out), (off, len(out))
        return steps, final

    # ---- the oracle running the reference's loop ------------------------------------------
    def run_oracle(self, oracle, cells, v_samp, const_k, beta, steps_env, wp_host):
        """Steps the oracle cells through the WRF steps; steps_env: the environment the Fortran
        side derived from the fields each step (its dump).  Returns per step the particle counts."""
        from wrf_partmc_amd.data import zhang_surface_parameters
        nx, nz, ny, Cn, n = self.nx, self.nz, self.ny, self.n_class, self.n_cells
        cell = lambda i, k, j: (j * nz + k) * nx + i
        seed = self.seed
        gas = self.gas0.copy()
        kcache = {}
        old_height = None
        counts_per_step = []
        level = lambda times, t: max(1, min(int(np.searchsorted(times, t, side="right")), len(times))) - 1
        fake = type("Fake", (), {})()
        for c, oc in enumerate(cells):       # grid%allow_doubling, grid%allow_halving (both .true.) in every add
            oc.set_weight_preparation(self.n_ideal[c], True, True)
        for s in range(self.n_steps):
            env, g, t = steps_env[s], self.G[s], float(s) * self.dt
            wrf_step, chem = int(t / self.dt), (math.fmod(t, self.chem_dt) == 0.0)
            height = env["height"]
            h_old = height if old_height is None else old_height
            # ---- partmc_timestep (driver:169-254) over the cells of its loop nest
            for c, oc in enumerate(cells):
                oc.temp, oc.pressure = float(env["temp"][c]), float(env["pres"][c])
                if not self.inside[c]:
                    continue
                if chem:
                    key = (oc.temp, oc.pressure)
                    if key not in kcache:
                        kcache[key] = oracle.kmax_table(K_BROWN, v_samp, self.density.min(), self.density.max(),
                                                        oc.temp, oc.pressure, const_k, beta)
                    oc.kmax[:] = kcache[key]
                    st, ns, nc = oc.mc_coag(K_BROWN, self.chem_dt, seed, c, int(t / self.chem_dt), const_k, beta)
                    assert st == 0, (s, c, st)
                e = sc = None
                if self.has_em[c]:
                    e = oracle.interp_1d(self.em_time, self.gas_em[c], t)
                    sc = oracle.interp_1d(self.em_time, self.em_scale, t)
                b = oracle.interp_1d(self.bg_time, self.gas_bg[c], t)
                rt = oracle.interp_1d(self.bg_time, self.gas_dil[c], t)
                gas[c] = oracle.scenario_update_gas(gas[c], self.dt, oc.temp, oc.pressure, float(height[c]),
                                                    float(h_old[c]), e, sc or 0.0, b, rt)
                oe = None
                if self.has_em[c]:
                    p = level(self.em_time, t)
                    oe = [dict(m, num_conc=m["num_conc"] * (self.em_scale[p] / float(height[c])))
                          for m in self.em[c][p]]
                p = level(self.bg_time, t)
                _, oc.next_id, _, _, _ = oc.scenario_update(self.dt, t, seed, c, wrf_step, oe, float(self.bg_dil[c, p]),
                                                            self.bg[c][p], oc.next_id, float(height[c]), float(h_old[c]))
                if self.sea_salt_opt and self.lev[c] == 0 and g["xland"][self.jj[c] + 1, self.ii[c] + 1] > 1.5:
                    u10, v10 = g["u10"][self.jj[c] + 1, self.ii[c] + 1], g["v10"][self.jj[c] + 1, self.ii[c] + 1]
                    wind = float(np.sqrt(F32(u10 * u10 + v10 * v10)))
                    st, oc.next_id, _ = oc.seasalt_emissions(self.dt, t, seed, c, wrf_step, self.sea_salt_opt, wind,
                                                             float(height[c]), self.i_na - 1, self.i_cl - 1,
                                                             [m if m < self.n_class else 0 for m in range(5)], [0, 1, 2, 0, 0],
                                                             oc.next_id)
                    assert st == 0
                st, oc.next_id = oc.rebalance(self.n_ideal[c], True, True, seed, c, wrf_step, oc.next_id)
                assert st == 0
            # ---- wrf_pmc_trans_driver (wrf_pmc_trans.F90:37-200)
            cnts = np.array([[int((oc.wclass[:int(oc.n_part[0])] == q).sum()) for q in range(Cn)] for oc in cells])
            mag = np.array([oc.mag for oc in cells])
            ph, pv, nsub, bad = oracle.transport_probs(nx, nz, ny, Cn, self.dt, self.dx, self.dy, self.F[s], cnts, mag,
                                                       do_vertmix=self.vertmix, all_columns=self.periodic,
                                                       i_offset=0, j_offset=0, global_nx=nx, global_ny=ny)
            assert not bad.any(), (s, bad)
            cv = env["cell_volume"]
            next_id = np.array([oc.next_id for oc in cells], dtype=np.int64)
            oracle.transport_preweight(cells, nx, nz, ny, ph, pv, cv, self.n_ideal, seed, 0, wrf_step,
                                       self.periodic, self.periodic)
            oracle.set_transport_law(1)          # (the driver's choice: wrf_pmc_trans_driver_pmcx)
            try:
                st, nm, na = oracle.transport(cells, nx, nz, ny, ph, pv, cv, seed, 0, wrf_step, next_id,
                                              self.periodic, self.periodic, lateral=not self.periodic)
            finally:
                oracle.set_transport_law()
            assert st == 0
            for c, oc in enumerate(cells):
                oc.next_id = int(next_id[c])
                if self.inside[c]:
                    st, oc.next_id = oc.rebalance(self.n_ideal[c], True, True, seed, c, wrf_step | (1 << 30), oc.next_id)
                    assert st == 0
            if not self.periodic:
                # the refill of the boundary cells: the host logic in its Python form (an independent
                # statement of wrf_pmc_trans_aero.F90:803-955) names the cells, the oracle acts
                u_w, u_e = g["u_2"][1:ny + 1, :nz, 1], g["u_2"][1:ny + 1, :nz, nx + 1]
                v_s, v_n = g["v_2"][1, :nz, 1:nx + 1].T, g["v_2"][ny + 1, :nz, 1:nx + 1].T
                u_wo, u_eo = g["u_1"][1:ny + 1, :nz, 1], g["u_1"][1:ny + 1, :nz, nx + 1]
                v_so, v_no = g["v_1"][1, :nz, 1:nx + 1].T, g["v_1"][ny + 1, :nz, 1:nx + 1].T
                p_t, p_tm1 = [int(np.searchsorted(self.bg_time, x, side="right")) for x in (t, t - self.dt)]
                change = p_t > p_tm1 and p_t > 1
                fake.cfg = type("C", (), dict(n_cells=n))()
                lists = {}
                fake.boundary_copy = lambda f, t_: lists.update(copy=(list(f), list(t_)))
                fake.boundary_resample = lambda cl, thr, *a: (lists.update(res=(list(cl), list(thr))) or
                                                              (np.zeros(len(cl), np.int32), np.zeros(len(cl), np.int32)))
                wp_host.CellBlock.lateral_boundary_conditions(
                    fake, nx, nz, ny, u_w, u_e, v_s, v_n, [dict()] * self.n_mode_b, 0.0, 1.0, seed=seed, step=wrf_step,
                    u_west_old=u_wo, u_east_old=u_eo, v_south_old=v_so, v_north_old=v_no,
                    change_boundary_conditions=change)
                for f_, t_ in zip(*lists.get("copy", ([], []))):
                    cells[t_].next_id = cells[t_].boundary_copy_from(cells[f_], cells[t_].next_id)
                p = level(self.bg_time, t)
                for c, thr in zip(*lists.get("res", ([], []))):
                    st, cells[c].next_id, _, _ = cells[c].boundary_resample(
                        thr, self.bg[c][p], 1.0 / float(env["inv_density"][c]), 0.0, seed, c, wrf_step, cells[c].next_id)
                    assert st == 0, (s, c, st, int(cells[c].n_part[0]), cells[c].cap)
            # dry deposition on the lowest level of the interior columns
            for j in range(1, ny - 1):
                for i in range(1, nx - 1):
                    c = cell(i, 0, j)
                    alpha, big_a = zhang_surface_parameters(int(g["ivgtyp"][j + 1, i + 1]))
                    cells[c].dry_deposition(self.dt, seed, c, wrf_step, float(env["aer_res"][c]),
                                            float(F32(g["ust"][j + 1, i + 1])), float(alpha[0]), float(big_a[0]),
                                            float(height[c]), float(env["inv_density"][c]))
            old_height = height
            counts_per_step.append(np.array([int(oc.n_part[0]) for oc in cells]))
        return counts_per_step, gas
