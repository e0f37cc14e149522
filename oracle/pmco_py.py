"""ctypes binding of the CPU oracle (oracle/libpmco.so).

ORACLE -- test infrastructure, not product code.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KERNEL_SEDI, KERNEL_ADDITIVE, KERNEL_CONSTANT, KERNEL_BROWN, KERNEL_ZERO = 1, 2, 3, 4, 5
KERNEL_BROWN_SEDI = 101

ST_EVENT_OVERFLOW, ST_APPEND_OVERFLOW, ST_KMAX_EXCEEDED, ST_OUT_OF_GRID, ST_CAP_OVERFLOW = 1, 2, 4, 8, 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class KernelParams(C.Structure):
    _fields_ = [("const_k", C.c_double), ("additive_beta", C.c_double)]


class CellStruct(C.Structure):
    _fields_ = [
        ("n_spec", C.c_int32), ("n_class", C.c_int32), ("n_source", C.c_int32),
        ("cap", C.c_int32), ("n_bin", C.c_int32),
        ("max_events", C.c_int32), ("max_appends", C.c_int32),
        ("density", _dp), ("v_edge", _dp), ("kmax", _dp),
        ("temp", C.c_double), ("pressure", C.c_double),
        ("mag", _dp), ("n_part", _ip), ("vol", _dp), ("vtot", _dp), ("mass", _dp),
        ("wclass", _ip), ("id", _lp), ("lct", _dp), ("gct", _dp), ("n_orig", _ip),
        ("pcap", C.c_int32), ("sort_valid", _ip), ("perm", _ip), ("gcnt", _ip), ("goff", _ip),
        ("n_comp_max", C.c_int32), ("comp_n", _ip), ("comp_src", _ip), ("comp_time", _dp),
        ("prep_n_ideal", _dp), ("prep_doubling", C.c_int32), ("prep_halving", C.c_int32),
    ]


class AeroDist(C.Structure):
    _fields_ = [
        ("n_mode", C.c_int32), ("num_conc", _dp), ("ln_gm_radius", _dp), ("ln_gsd", _dp),
        ("vol_frac", _dp), ("mode_class", _ip), ("mode_source", _ip),
    ]


class GenSpec(C.Structure):
    _fields_ = [
        ("n_mode", C.c_int32), ("mode_frac", _dp), ("ln_gm_radius", _dp),
        ("ln_gsd", _dp), ("vol_frac", _dp), ("mode_class", _ip), ("mode_source", _ip),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libpmco.so")
    srcs = [os.path.join(_HERE, f) for f in ("pmco.c", "pmco.h", "pmco_math.h")]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def build_native():
    """The -O3 -march=native build for bench.py's cpu_baseline, compiled on this host."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
    return os.path.join(_HERE, "_native", "libpmco.so")


LN_10 = 2.302585092994046   # ln 10, correctly rounded


def det_log(x):
    """The logarithm of a mode's parameters: pmco_log, not numpy's nor libm's (numpy's vectorised
    log and glibc's differ in the last bit for one argument in a few hundred; the product forms
    these on the host with the same operations, pmcx_device_math.h)."""
    return lib().pmco_log_(float(x))


def lib():
    global _LIB
    if _LIB is None:
        # PMCO_LIBRARY: bench.py points its timing workers at the native build
        L = C.CDLL(os.environ.get("PMCO_LIBRARY") or build())
        L.pmco_exp_.restype = C.c_double; L.pmco_exp_.argtypes = [C.c_double]
        L.pmco_log_.restype = C.c_double; L.pmco_log_.argtypes = [C.c_double]
        L.pmco_cbrt_.restype = C.c_double; L.pmco_cbrt_.argtypes = [C.c_double]
        L.pmco_norminv_.restype = C.c_double; L.pmco_norminv_.argtypes = [C.c_double]
        L.pmco_poisson_.restype = C.c_int32; L.pmco_poisson_.argtypes = [C.c_double, C.c_double]
        L.pmco_kernel.restype = C.c_double
        L.pmco_kernel.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.POINTER(KernelParams)]
        L.pmco_bin_of.restype = C.c_int
        L.pmco_bin_of.argtypes = [_dp, C.c_int, C.c_double]
        L.pmco_mc_coag.restype = C.c_int
        L.pmco_mc_coag.argtypes = [C.POINTER(CellStruct), C.c_int, C.POINTER(KernelParams),
                                   C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, _ip, _ip]
        L.pmco_rebalance.restype = C.c_int
        L.pmco_rebalance.argtypes = [C.POINTER(CellStruct), _dp, C.c_int, C.c_int, C.c_uint64,
                                     C.c_uint32, C.c_uint32, _lp]
        L.pmco_generate.restype = None
        L.pmco_generate.argtypes = [C.POINTER(CellStruct), C.POINTER(GenSpec), C.c_int32,
                                    C.c_uint64, C.c_uint32, C.c_int64, C.c_double]
        L.pmco_scenario_update.restype = C.c_int
        L.pmco_scenario_update.argtypes = [
            C.POINTER(CellStruct), C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32,
            C.POINTER(AeroDist), C.c_double, C.POINTER(AeroDist), _lp, _ip, _ip, _ip]
        L.pmco_scenario_update_h.restype = C.c_int
        L.pmco_scenario_update_h.argtypes = L.pmco_scenario_update.argtypes + [C.c_double, C.c_double]
        L.pmco_condense.restype = None
        L.pmco_condense.argtypes = [C.POINTER(CellStruct), _dp, C.c_int, C.c_double, C.c_double,
                                    C.c_int]
        L.pmco_refresh_caches.restype = None
        L.pmco_refresh_caches.argtypes = [C.POINTER(CellStruct)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    lib().pmco_philox(_p(c, u32p), _p(k, u32p), _p(o, u32p))
    return o


def kernel(ktype, v1, m1, v2, m2, temp, pressure, const_k=0.0, additive_beta=0.0):
    kp = KernelParams(const_k, additive_beta)
    return lib().pmco_kernel(ktype, v1, m1, v2, m2, temp, pressure, C.byref(kp))


def make_bin_tables(n_bin, r_min, r_max):
    v_edge = np.zeros(n_bin + 1)
    v_samp = np.zeros(n_bin * 3)
    lib().pmco_make_bin_tables(C.c_int(n_bin), C.c_double(r_min), C.c_double(r_max),
                               _p(v_edge, _dp), _p(v_samp, _dp))
    return v_edge, v_samp


def dep_velocity(diameter, density, temp, rrho, ustar, ra, alpha, big_a):
    f = lib().pmco_dep_velocity_
    f.restype = C.c_double
    f.argtypes = [C.c_double] * 8
    return f(diameter, density, temp, rrho, ustar, ra, alpha, big_a)


def kmax_table(ktype, v_samp, rho_min, rho_max, temp, pressure, const_k=0.0, additive_beta=0.0):
    n_bin = len(v_samp) // 3
    out = np.zeros(n_bin * n_bin)
    kp = KernelParams(const_k, additive_beta)
    lib().pmco_kmax_table(C.c_int(ktype), C.c_int(n_bin), _p(np.ascontiguousarray(v_samp), _dp),
                          C.c_double(rho_min), C.c_double(rho_max), C.c_double(temp),
                          C.c_double(pressure), C.byref(kp), _p(out, _dp))
    return out.reshape(n_bin, n_bin)


class Cell:
    """One cell's state as numpy arrays + the C view over them."""

    def __init__(self, n_spec, n_class, n_source, cap, density, v_edge, temp=288.15,
                 pressure=101325.0, max_events=1 << 20, max_appends=1 << 20, n_comp_max=0):
        self.n_spec, self.n_class, self.n_source, self.cap = n_spec, n_class, n_source, cap
        # particle%component(:): the first n_comp_max (source, create_time) pairs of each particle
        self.n_comp_max = n_comp_max
        self.prep_n_ideal, self.prep_doubling, self.prep_halving = None, True, True
        self.comp_n = np.zeros(cap, dtype=np.int32)
        self.comp_src = np.full((cap, max(n_comp_max, 1)), -1, dtype=np.int32)
        self.comp_time = np.zeros((cap, max(n_comp_max, 1)))
        self.density = np.ascontiguousarray(density, dtype=np.float64)
        self.v_edge = np.ascontiguousarray(v_edge, dtype=np.float64)
        self.n_bin = len(self.v_edge) - 1
        self.kmax = np.zeros((self.n_bin, self.n_bin))
        self.temp, self.pressure = float(temp), float(pressure)
        self.max_events, self.max_appends = max_events, max_appends
        self.mag = np.zeros(n_class)
        self.n_part = np.zeros(1, dtype=np.int32)
        self.vol = np.zeros((n_spec, cap))
        self.vtot = np.zeros(cap)
        self.mass = np.zeros(cap)
        self.wclass = np.zeros(cap, dtype=np.int32)
        self.id = np.zeros(cap, dtype=np.int64)
        self.lct = np.zeros(cap)
        self.gct = np.zeros(cap)
        self.n_orig = np.zeros((max(n_source, 1), cap), dtype=np.int32)
        # aero_sorted: kept between calls while sort_valid (set it to 0 after
        # editing the particle arrays by hand)
        G = n_class * self.n_bin
        # (16-bit offsets on the device: the lists' layout adapts to what perm holds, pmco_sort)
        self.pcap = min(65535, cap + cap // 8 + min(16 * G, 8192))
        self.sort_valid = np.zeros(1, dtype=np.int32)
        self.perm = np.zeros(self.pcap, dtype=np.int32)
        self.gcnt = np.zeros(G, dtype=np.int32)
        self.goff = np.zeros(G + 1, dtype=np.int32)

    def view(self):
        s = CellStruct()
        s.n_spec, s.n_class, s.n_source = self.n_spec, self.n_class, self.n_source
        s.cap, s.n_bin = self.cap, self.n_bin
        s.max_events, s.max_appends = self.max_events, self.max_appends
        s.density = _p(self.density, _dp); s.v_edge = _p(self.v_edge, _dp)
        s.kmax = _p(self.kmax, _dp)
        s.temp, s.pressure = self.temp, self.pressure
        s.mag = _p(self.mag, _dp); s.n_part = _p(self.n_part, _ip)
        s.vol = _p(self.vol, _dp); s.vtot = _p(self.vtot, _dp); s.mass = _p(self.mass, _dp)
        s.wclass = _p(self.wclass, _ip); s.id = _p(self.id, _lp)
        s.lct = _p(self.lct, _dp); s.gct = _p(self.gct, _dp); s.n_orig = _p(self.n_orig, _ip)
        s.pcap = self.pcap; s.sort_valid = _p(self.sort_valid, _ip)
        s.perm = _p(self.perm, _ip); s.gcnt = _p(self.gcnt, _ip); s.goff = _p(self.goff, _ip)
        s.n_comp_max = self.n_comp_max
        s.comp_n = _p(self.comp_n, _ip); s.comp_src = _p(self.comp_src, _ip)
        s.comp_time = _p(self.comp_time, _dp)
        if self.prep_n_ideal is not None:
            s.prep_n_ideal = _p(self.prep_n_ideal, _dp)
            s.prep_doubling, s.prep_halving = int(self.prep_doubling), int(self.prep_halving)
        return s

    def set_weight_preparation(self, n_ideal, allow_doubling=True, allow_halving=True):
        """aero_state_prepare_weight_for_add in every add of particles from a distribution
        (emission, background, sea salt, the refill of an emptied boundary cell); n_ideal None: off."""
        self.prep_n_ideal = None if n_ideal is None else np.ascontiguousarray(n_ideal, dtype=np.float64).copy()
        self.prep_doubling, self.prep_halving = bool(allow_doubling), bool(allow_halving)

    def refresh_caches(self):
        lib().pmco_refresh_caches(C.byref(self.view()))

    def mc_coag(self, ktype, del_t, seed, cell_id, step, const_k=0.0, additive_beta=0.0):
        kp = KernelParams(const_k, additive_beta)
        ns, nc = C.c_int32(0), C.c_int32(0)
        st = lib().pmco_mc_coag(C.byref(self.view()), ktype, C.byref(kp), del_t, seed, cell_id,
                                step, C.byref(ns), C.byref(nc))
        return st, ns.value, nc.value

    def rebalance(self, n_ideal, allow_doubling, allow_halving, seed, cell_id, step, next_id):
        ni = np.ascontiguousarray(n_ideal, dtype=np.float64)
        nid = C.c_int64(next_id)
        st = lib().pmco_rebalance(C.byref(self.view()), _p(ni, _dp), int(allow_doubling),
                                  int(allow_halving), seed, cell_id, step, C.byref(nid))
        return st, nid.value

    @staticmethod
    def _dist(modes):
        if not modes:
            return None, None
        keep = dict(
            nc=np.array([m["num_conc"] for m in modes], dtype=np.float64),
            lg=np.array([det_log(m["gm_radius"]) for m in modes], dtype=np.float64),
            ls=np.array([m["gsd_log10"] * LN_10 for m in modes], dtype=np.float64),
            vf=np.ascontiguousarray([m["vol_frac"] for m in modes], dtype=np.float64),
            mc=np.array([m.get("wclass", 0) for m in modes], dtype=np.int32),
            ms=np.array([m.get("source", 0) for m in modes], dtype=np.int32))
        d = AeroDist(len(modes), _p(keep["nc"], _dp), _p(keep["lg"], _dp), _p(keep["ls"], _dp),
                     _p(keep["vf"], _dp), _p(keep["mc"], _ip), _p(keep["ms"], _ip))
        return d, keep

    def scenario_update(self, dt, create_time, seed, cell_id, step, emission, dilution_rate,
                        background, next_id, height=0.0, old_height=0.0):
        """emission/background: lists of dict(num_conc, gm_radius, gsd_log10, vol_frac, wclass, source).
        height, old_height: env_state%height now and a step ago (0: unchanged)."""
        e, ke = self._dist(emission)
        b, kb = self._dist(background)
        nid = C.c_int64(next_id)
        ne, ni, no = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        st = lib().pmco_scenario_update_h(C.byref(self.view()), dt, create_time, seed, cell_id, step,
                                          C.byref(e) if e else None, dilution_rate,
                                          C.byref(b) if b else None, C.byref(nid), C.byref(ne),
                                          C.byref(ni), C.byref(no), float(height), float(old_height))
        return st, nid.value, ne.value, ni.value, no.value

    def boundary_copy_from(self, src, next_id):
        """pmco_boundary_copy: this cell becomes a copy of src; returns the new next_id."""
        nid = C.c_int64(next_id)
        f = lib().pmco_boundary_copy
        f.restype = None
        f.argtypes = [C.POINTER(CellStruct), C.POINTER(CellStruct), _lp]
        f(C.byref(self.view()), C.byref(src.view()), C.byref(nid))
        return nid.value

    def boundary_resample(self, throw_away, background, density, create_time, seed, cell_id, step,
                          next_id):
        """pmco_boundary_resample; background: list of mode dicts (num_conc = mixing ratio).
        Returns (status, next_id, n_added, n_removed)."""
        b, kb = self._dist(background)
        nid = C.c_int64(next_id)
        na, nr = C.c_int32(0), C.c_int32(0)
        f = lib().pmco_boundary_resample
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CellStruct), C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_uint64,
                      C.c_uint32, C.c_uint32, _lp, _ip, _ip]
        st = f(C.byref(self.view()), int(throw_away), C.cast(C.byref(b), C.c_void_p), density,
               create_time, seed, cell_id, step, C.byref(nid), C.byref(na), C.byref(nr))
        return st, nid.value, na.value, nr.value

    def seasalt_emissions(self, dt, create_time, seed, cell_id, step, sea_salt_opt, wind_speed,
                          height, i_na, i_cl, mode_class, mode_source, next_id):
        """pmco_seasalt_emissions of this cell; returns (status, next_id, n_emit)."""
        f = lib().pmco_seasalt_emissions
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CellStruct), C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                      C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _ip, _ip, _lp, _ip]
        mc = np.ascontiguousarray(mode_class, dtype=np.int32)
        ms = np.ascontiguousarray(mode_source, dtype=np.int32)
        nid, ne = C.c_int64(next_id), C.c_int32(0)
        st = f(C.byref(self.view()), dt, create_time, seed, cell_id, step, sea_salt_opt, wind_speed,
               height, i_na, i_cl, _p(mc, _ip), _p(ms, _ip), C.byref(nid), C.byref(ne))
        return st, nid.value, ne.value

    def dry_deposition(self, dt, seed, cell_id, step, aer_res_a, ustar, alpha, big_a, height,
                       inv_density):
        """dry_dep_aero_state of this cell; returns (n_removed, unrealistic)."""
        f = lib().pmco_dry_deposition
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CellStruct), C.c_double, C.c_uint64, C.c_uint32, C.c_uint32] + \
                     [C.c_double] * 6 + [C.POINTER(C.c_int32)]
        bad = C.c_int32(0)
        nd = f(C.byref(self.view()), dt, seed, cell_id, step, aer_res_a, ustar, alpha, big_a, height,
               inv_density, C.byref(bad))
        return nd, bad.value

    def diagnostics_ext(self, i_water):
        sm, sn, pm = np.zeros(self.n_spec), np.zeros(max(self.n_source, 1)), np.zeros(3)
        f = lib().pmco_diagnostics_ext
        f.restype = None
        f.argtypes = [C.POINTER(CellStruct), C.c_int, _dp, _dp, _dp]
        f(C.byref(self.view()), int(i_water), _p(sm, _dp), _p(sn, _dp), _p(pm, _dp))
        return sm, sn[:self.n_source], pm

    def condense(self, kappa, i_water, rel_humid, del_t, n_sub=4):
        k = np.ascontiguousarray(kappa, dtype=np.float64)
        lib().pmco_condense(C.byref(self.view()), _p(k, _dp), int(i_water), float(rel_humid),
                            float(del_t), int(n_sub))

    def diagnostics(self, i_water, d_edge, d_width):
        n = len(d_width)
        tn, tm = C.c_double(0), C.c_double(0)
        nd, md = np.zeros(n), np.zeros(n)
        de = np.ascontiguousarray(d_edge, dtype=np.float64)
        dw = np.ascontiguousarray(d_width, dtype=np.float64)
        lib().pmco_diagnostics(C.byref(self.view()), C.c_int(i_water), C.c_int(n), _p(de, _dp),
                               _p(dw, _dp), C.byref(tn), C.byref(tm), _p(nd, _dp), _p(md, _dp))
        return tn.value, tm.value, nd, md

    def generate(self, modes, n, seed, cell_id, first_id=0, create_time=0.0):
        """modes: list of dict(frac, gm_radius, gsd_log10, vol_frac[n_spec], wclass, source)."""
        fr = np.cumsum([m["frac"] for m in modes], dtype=np.float64)
        fr = fr / fr[-1]
        lg = np.array([det_log(m["gm_radius"]) for m in modes], dtype=np.float64)
        ls = np.array([m["gsd_log10"] * LN_10 for m in modes], dtype=np.float64)
        vf = np.ascontiguousarray([m["vol_frac"] for m in modes], dtype=np.float64)
        mc = np.array([m.get("wclass", 0) for m in modes], dtype=np.int32)
        msrc = np.array([m.get("source", 0) for m in modes], dtype=np.int32)
        g = GenSpec(len(modes), _p(fr, _dp), _p(lg, _dp), _p(ls, _dp), _p(vf, _dp),
                    _p(mc, _ip), _p(msrc, _ip))
        lib().pmco_generate(C.byref(self.view()), C.byref(g), n, seed, cell_id, first_id,
                            create_time)
        return fr, lg, ls, vf, mc, msrc


def interp_1d(x, y, xq):
    """pmco_interp_1d on a time series y[n] (or y[n][m], every column)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    f = lib().pmco_interp_1d
    f.restype = C.c_double
    f.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_double]
    if y.ndim == 1:
        return f(len(x), _p(x, _dp), _p(y, _dp), 1, xq)
    m = y.shape[1]
    out = np.zeros(m)
    for j in range(m):
        col = y[:, j:].reshape(-1)
        out[j] = f(len(x), _p(x, _dp), C.cast(C.c_void_p(y.ctypes.data + 8 * j), _dp), m, xq)
    return out


def scenario_update_gas(mix_rat, dt, temp, pressure, height, old_height, emission,
                        emission_rate_scale, background, dilution_rate):
    """pmco_scenario_update_gas on one cell's mix_rat [n_gas] (in place; returns it)."""
    mr = np.ascontiguousarray(mix_rat, dtype=np.float64)
    em = None if emission is None else np.ascontiguousarray(emission, dtype=np.float64)
    bg = None if background is None else np.ascontiguousarray(background, dtype=np.float64)
    f = lib().pmco_scenario_update_gas
    f.restype = None
    f.argtypes = [C.c_int, _dp] + [C.c_double] * 5 + [_dp, C.c_double, _dp, C.c_double]
    f(len(mr), _p(mr, _dp), dt, temp, pressure, height, old_height,
      None if em is None else _p(em, _dp), emission_rate_scale,
      None if bg is None else _p(bg, _dp), dilution_rate)
    return mr


def seasalt_modes(sea_salt_opt, wind_speed):
    """(flux, gm_radius, gsd_log10, rate_scale, vol_frac_na) of the sea-salt source."""
    f = lib().pmco_seasalt_modes
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
    fl, gm, gs = np.zeros(5), np.zeros(5), np.zeros(5)
    sc, na = C.c_double(), C.c_double()
    m = f(sea_salt_opt, wind_speed, _p(fl, _dp), _p(gm, _dp), _p(gs, _dp), C.byref(sc), C.byref(na))
    return fl[:m], gm[:m], gs[:m], sc.value, na.value


def sort_slack(n, level):
    """Room behind a list of n entries at a level of the layout's ladder (pmco_sort_slack)."""
    f = lib().pmco_sort_slack
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int]
    return f(int(n), int(level))


def sort_level(n_part, s3, populated, n_groups, pcap):
    """The level of the ladder a cell's lists are laid out at (pmco_sort_level)."""
    f = lib().pmco_sort_level
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 5
    return f(int(n_part), int(s3), int(populated), int(n_groups), int(pcap))


def sort(cell):
    """aero_state_sort of one Cell; returns the status bits."""
    st = C.c_int(0)
    f = lib().pmco_sort
    f.restype = None
    f.argtypes = [C.POINTER(CellStruct), C.POINTER(C.c_int)]
    f(C.byref(cell.view()), C.byref(st))
    return st.value


class TransFields(C.Structure):
    _fields_ = [("nx", C.c_int), ("nz", C.c_int), ("ny", C.c_int), ("n_class", C.c_int),
                ("dt", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
                ("do_vertmix", C.c_int), ("all_columns", C.c_int), ("i_offset", C.c_int),
                ("j_offset", C.c_int), ("global_nx", C.c_int), ("global_ny", C.c_int)] + \
               [(n, C.POINTER(C.c_float)) for n in ("alt", "xkhh", "u_flux", "v_flux", "w_flux", "z",
                                                    "z_at_w", "exch_h")]


TRANS_FIELD_SHAPES = {   # (nx, nz, ny, n_class) -> shape
    "alt": lambda nx, nz, ny, C_: (ny + 2, nz, nx + 2), "xkhh": lambda nx, nz, ny, C_: (ny + 2, nz, nx + 2),
    "u_flux": lambda nx, nz, ny, C_: (C_, ny, nz, nx + 1), "v_flux": lambda nx, nz, ny, C_: (C_, ny + 1, nz, nx),
    "w_flux": lambda nx, nz, ny, C_: (C_, ny, nz + 1, nx), "z": lambda nx, nz, ny, C_: (ny, nz, nx),
    "z_at_w": lambda nx, nz, ny, C_: (ny, nz + 1, nx), "exch_h": lambda nx, nz, ny, C_: (ny, nz + 1, nx)}


def transport_probs(nx, nz, ny, n_class, dt, dx, dy, fields, n_parts, mag, do_vertmix=1,
                    all_columns=True, i_offset=0, j_offset=0, global_nx=None, global_ny=None):
    """pmco_transport_probs; fields: dict of float32 arrays (TRANS_FIELD_SHAPES).
    Returns (prob_h [n_cells][n_class][4], prob_v [n_cells][n_class][nz], n_substeps [ny][nx], bad)."""
    keep = {}
    tf = TransFields(nx, nz, ny, n_class, dt, dx, dy, int(do_vertmix), int(all_columns), i_offset,
                     j_offset, nx if global_nx is None else global_nx, ny if global_ny is None else global_ny)
    for name, shp in TRANS_FIELD_SHAPES.items():
        a = np.ascontiguousarray(fields[name], dtype=np.float32)
        assert a.shape == shp(nx, nz, ny, n_class), (name, a.shape)
        keep[name] = a
        setattr(tf, name, a.ctypes.data_as(C.POINTER(C.c_float)))
    n = nx * nz * ny
    npart = np.ascontiguousarray(n_parts, dtype=np.int32).reshape(n, n_class)
    mg = np.ascontiguousarray(mag, dtype=np.float64).reshape(n, n_class)
    ph, pv = np.zeros((n, n_class, 4)), np.zeros((n, n_class, nz))
    ns, bad = np.zeros((ny, nx), dtype=np.int32), np.zeros(n, dtype=np.int32)
    f = lib().pmco_transport_probs
    f.restype = C.c_int
    f.argtypes = [C.POINTER(TransFields), _ip, _dp, _dp, _dp, _ip, _ip]
    rc = f(C.byref(tf), _p(npart, _ip), _p(mg, _dp), _p(ph, _dp), _p(pv, _dp), _p(ns, _ip), _p(bad, _ip))
    if rc:
        raise ValueError("pmco_transport_probs: bad dimensions")
    return ph, pv, ns, bad


def transport_preweight(cells, nx, nz, ny, prob_h, prob_v, cell_volume, n_ideal, seed, cell_id_base,
                        step, periodic_x=True, periodic_y=True):
    """pmco_transport_preweight over a list of Cell objects; returns (expected [n][C], n_removed)."""
    n = len(cells)
    Cn = cells[0].n_class
    arr = (CellStruct * n)(*[c.view() for c in cells])
    ph = np.ascontiguousarray(prob_h, dtype=np.float64)
    pv = np.ascontiguousarray(prob_v, dtype=np.float64)
    cv = np.ascontiguousarray(cell_volume, dtype=np.float64)
    ni = np.ascontiguousarray(np.broadcast_to(np.asarray(n_ideal, dtype=np.float64), (n, Cn)))
    ex, nr = np.zeros((n, Cn)), np.zeros(n, dtype=np.int32)
    f = lib().pmco_transport_preweight
    f.restype = C.c_int
    f.argtypes = [C.POINTER(CellStruct), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                  _dp, _dp, _dp, C.c_uint64, C.c_uint32, C.c_uint32, _dp, _ip]
    f(arr, n, nx, nz, ny, int(periodic_x), int(periodic_y), _p(ph, _dp), _p(pv, _dp), _p(cv, _dp),
      _p(ni, _dp), seed, cell_id_base, step, _p(ex, _dp), _p(nr, _ip))
    return ex, nr


DEFAULT_TRANSPORT_LAW = 1


def set_transport_law(law=DEFAULT_TRANSPORT_LAW):
    """1 (default): the reference's r < 1 law -- movers arrive once, particles that stay emit the
    other copies; 0: a mover carries 1 + geometric copies (pmcx_set_transport_law, pmco.c)."""
    lib().pmco_set_transport_law(int(law))


def transport(cells, nx, nz, ny, prob_h, prob_v, cell_volume, seed, cell_id_base, step, next_id,
              periodic_x=True, periodic_y=True, lateral=False):
    """pmco_transport over a list of Cell objects (WRF order).  next_id: int64 array, updated.
    lateral: pmco_transport_lateral (open lateral boundaries, neither direction periodic)."""
    n = len(cells)
    if lateral:
        assert not periodic_x and not periodic_y
        arr = (CellStruct * n)(*[c.view() for c in cells])
        ph = np.ascontiguousarray(prob_h, dtype=np.float64)
        pv = np.ascontiguousarray(prob_v, dtype=np.float64)
        cv = np.ascontiguousarray(cell_volume, dtype=np.float64)
        nm, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        f = lib().pmco_transport_lateral
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CellStruct), C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                      C.c_uint64, C.c_uint32, C.c_uint32, _lp, _ip, _ip]
        st = f(arr, n, nx, nz, ny, _p(ph, _dp), _p(pv, _dp), _p(cv, _dp), seed, cell_id_base, step,
               _p(next_id, _lp), _p(nm, _ip), _p(na, _ip))
        return st, nm, na
    arr = (CellStruct * n)(*[c.view() for c in cells])
    ph = np.ascontiguousarray(prob_h, dtype=np.float64)
    pv = np.ascontiguousarray(prob_v, dtype=np.float64)
    cv = np.ascontiguousarray(cell_volume, dtype=np.float64)
    nm, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    f = lib().pmco_transport
    f.restype = C.c_int
    f.argtypes = [C.POINTER(CellStruct), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                  _dp, _dp, C.c_uint64, C.c_uint32, C.c_uint32, _lp, _ip, _ip]
    st = f(arr, n, nx, nz, ny, int(periodic_x), int(periodic_y), _p(ph, _dp), _p(pv, _dp),
           _p(cv, _dp), seed, cell_id_base, step, _p(next_id, _lp), _p(nm, _ip), _p(na, _ip))
    return st, nm, na
