"""CPU tests (-m "not gpu"): pin the oracle.

The reference holds no golden vectors for this path, so the
oracle is pinned by: Random123 known-answer vectors; libm/scipy for the
deterministic math; an independent numpy transcription of the in-tree Fuchs
Brownian form (WRFV3/chem/module_mosaic_coag.F:746-757,898-936) and textbook
values; analytic Smoluchowski solutions (constant and additive kernels);
exact conservation laws; and the committed fixtures in tests/golden/."""
import json
import math
import os

import numpy as np
import pytest

from tests import parity

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
K_BROWN, K_SEDI, K_ADD, K_CONST, K_ZERO, K_BS = 4, 1, 2, 3, 5, 101


# ------------------------------------------------------------------ RNG + math

def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in oracle.philox([0] * 4, [0] * 2)] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in oracle.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                          [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_deterministic_math_accuracy(oracle):
    from scipy.stats import norm
    L = oracle.lib()
    for x in np.concatenate([-np.logspace(-8, 2.85, 400), np.logspace(-8, 2.8, 100)]):
        assert abs(L.pmco_exp_(x) - math.exp(x)) <= 4e-16 * math.exp(x)
    for x in np.logspace(-300, 300, 900):
        assert abs(L.pmco_log_(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))
    for x in np.logspace(-40, 5, 900):
        y = L.pmco_cbrt_(x)
        assert abs(y * y * y - x) <= 2.5e-15 * x
    for p in np.concatenate([np.logspace(-15, -2, 60), np.linspace(0.01, 0.99, 99)]):
        assert abs(L.pmco_norminv_(p) - norm.ppf(p)) <= 2e-9 * max(1.0, abs(norm.ppf(p)))


def test_poisson_moments(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(3)
    for mean in (0.05, 0.7, 3.0, 25.0, 180.0, 450.0, 5000.0):
        x = np.array([L.pmco_poisson_(mean, u) for u in rng.uniform(0, 1, 40000)])
        assert abs(x.mean() - mean) < 5 * math.sqrt(mean / x.size) + 0.01 * (mean >= 200)
        assert abs(x.var() - mean) < 0.06 * mean + 0.01
    assert L.pmco_poisson_(0.0, 0.5) == 0 and L.pmco_poisson_(-1.0, 0.5) == 0


# ----------------------------------------------------------------- kernels

def fuchs_cgs(r1_cm, r2_cm, rho1, rho2, tk, patm, slip=(1.257, 0.42, 1.1)):
    """Independent transcription of the functional form of
    WRFV3/chem/module_mosaic_coag.F:746-757 (air properties) and :898-936
    (Fuchs Brownian coefficient), CGS units, returns cm^3/s."""
    boltg, wtair, avg, rgas2 = 1.38054e-16, 28.966, 6.02252e23, 0.08206
    amu = 1.8325e-4 * (296.16 + 120.0) / (tk + 120.0) * (tk / 296.16) ** 1.5
    rho3 = patm * wtair * 0.001 / (rgas2 * tk)
    vthermg = math.sqrt(8.0 * boltg * tk * avg / (math.pi * wtair))
    gmfp = 2.0 * (amu / rho3) / vthermg
    out = []
    for r, rho in ((r1_cm, rho1), (r2_cm, rho2)):
        kn = gmfp / r
        bpm = 1.0 + kn * (slip[0] + slip[1] * math.exp(-slip[2] / kn))
        dif = boltg * tk * bpm / (6.0 * math.pi * r * amu)
        vol = 4.0 / 3.0 * math.pi * r ** 3
        vth = math.sqrt(8.0 * boltg * tk / (math.pi * rho * vol))
        pmfp = 8.0 * dif / (math.pi * vth)
        delt = ((2 * r + pmfp) ** 3 - (4 * r * r + pmfp * pmfp) ** 1.5) / (6.0 * r * pmfp) - 2 * r
        out.append((dif, vth * vth, delt * delt))
    rs = r1_cm + r2_cm
    sd = out[0][0] + out[1][0]
    ggr = math.sqrt(out[0][1] + out[1][1])
    deltr = math.sqrt(out[0][2] + out[1][2])
    return 4.0 * math.pi * rs * sd / (rs / (rs + deltr) + 4.0 * sd / (rs * ggr))


def test_brownian_kernel_matches_in_tree_form(oracle):
    rng = np.random.RandomState(1)
    for T, p in ((288.15, 101325.0), (220.0, 25000.0)):
        for _ in range(200):
            d1, d2 = 10 ** rng.uniform(-9, -5, 2)
            rho1, rho2 = rng.uniform(900, 2700, 2)
            v1, v2 = math.pi / 6 * d1 ** 3, math.pi / 6 * d2 ** 3
            k = oracle.kernel(K_BROWN, v1, v1 * rho1, v2, v2 * rho2, T, p)
            same = fuchs_cgs(d1 * 50, d2 * 50, rho1 * 1e-3, rho2 * 1e-3, T, p / 101325.0,
                             slip=(1.249, 0.42, 0.87)) * 1e-6
            mosaic = fuchs_cgs(d1 * 50, d2 * 50, rho1 * 1e-3, rho2 * 1e-3, T, p / 101325.0) * 1e-6
            assert abs(k - same) <= 2e-4 * same      # constants differ in the 5th digit
            assert abs(k - mosaic) <= 0.05 * mosaic  # slip constants 1.257/0.42/1.1 in-tree


def test_brownian_kernel_textbook_values(oracle):
    # Seinfeld & Pandis, Table 13.3 (25 C, 1 atm, rho = 1 g/cm^3), cm^3/s
    table = {(0.01, 0.01): 18e-10, (0.1, 0.1): 14.4e-10, (1.0, 1.0): 6.9e-10,
             (0.01, 1.0): 3.4e-7}
    for (d1, d2), k_cgs in table.items():
        v1, v2 = math.pi / 6 * (d1 * 1e-6) ** 3, math.pi / 6 * (d2 * 1e-6) ** 3
        k = oracle.kernel(K_BROWN, v1, v1 * 1000, v2, v2 * 1000, 298.15, 101325.0)
        assert abs(k - k_cgs * 1e-6) <= 0.10 * k_cgs * 1e-6  # table rounding + slip constants


def test_kernel_limits_and_symmetry(oracle):
    # symmetric; free-molecular limit ~ sqrt(T); continuum limit of equal spheres 8kT/3mu
    v = lambda d: math.pi / 6 * d ** 3
    a = oracle.kernel(K_BROWN, v(3e-8), v(3e-8) * 1800, v(4e-7), v(4e-7) * 1200, 280.0, 9e4)
    b = oracle.kernel(K_BROWN, v(4e-7), v(4e-7) * 1200, v(3e-8), v(3e-8) * 1800, 280.0, 9e4)
    assert a == b
    T = 293.15
    mu = 1.8325e-5 * (296.16 + 120) / (T + 120) * (T / 296.16) ** 1.5
    k = oracle.kernel(K_BROWN, v(2e-5), v(2e-5) * 1000, v(2e-5), v(2e-5) * 1000, T, 101325.0)
    assert abs(k - 8 * 1.3806505e-23 * T / (3 * mu)) < 0.02 * k
    assert oracle.kernel(K_ZERO, 1e-21, 1e-18, 1e-21, 1e-18, T, 1e5) == 0.0
    assert oracle.kernel(K_ADD, 1e-21, 1e-18, 3e-21, 1e-18, T, 1e5, additive_beta=1000.0) == 1000.0 * 4e-21
    # sedimentation: zero between equal particles, positive and growing with size contrast
    assert oracle.kernel(K_SEDI, v(2e-6), v(2e-6) * 1000, v(2e-6), v(2e-6) * 1000, T, 1e5) == 0.0
    s1 = oracle.kernel(K_SEDI, v(2e-6), v(2e-6) * 1000, v(2e-5), v(2e-5) * 1000, T, 1e5)
    s2 = oracle.kernel(K_SEDI, v(2e-6), v(2e-6) * 1000, v(4e-5), v(4e-5) * 1000, T, 1e5)
    assert 0 < s1 < s2
    bs = oracle.kernel(K_BS, v(2e-6), v(2e-6) * 1000, v(2e-5), v(2e-5) * 1000, T, 1e5)
    br = oracle.kernel(K_BROWN, v(2e-6), v(2e-6) * 1000, v(2e-5), v(2e-5) * 1000, T, 1e5)
    assert bs == br + s1


def test_kmax_bounds_random_pairs(oracle):
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    rng = np.random.RandomState(2)
    for ktype in (K_BROWN, K_BS):
        km = oracle.kmax_table(ktype, vs, 1000.0, 2600.0, 275.0, 8e4)
        assert np.array_equal(km, km.T)
        worst = 0.0
        for _ in range(4000):
            b1, b2 = rng.randint(5, 60, 2)
            v1 = ve[b1] * (ve[b1 + 1] / ve[b1]) ** rng.uniform()
            v2 = ve[b2] * (ve[b2 + 1] / ve[b2]) ** rng.uniform()
            k = oracle.kernel(ktype, v1, v1 * rng.uniform(1000, 2600), v2,
                              v2 * rng.uniform(1000, 2600), 275.0, 8e4)
            worst = max(worst, k / km[b1, b2])
        # 3x3-point bound [PartMC est_k_minmax]: tight, may be exceeded by a few percent
        assert worst < 1.1


def test_bin_of(oracle):
    ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    L = oracle.lib()
    p = ve.ctypes.data_as(oracle._dp)
    for b in range(80):
        assert L.pmco_bin_of(p, 80, ve[b]) == b
        assert L.pmco_bin_of(p, 80, np.nextafter(ve[b + 1], 0)) == b
    assert L.pmco_bin_of(p, 80, ve[0] / 10) == 0 and L.pmco_bin_of(p, 80, ve[80] * 10) == 79


# ------------------------------------------------------- mc_coag: exact laws

def _cell(oracle, n_part, n_spec, n_class, kernel, seed, cell_id=0, coarse=False, T=288.15,
          p=101325.0, conc_scale=1.0, const_k=0.25 / (60 * 2e8), beta=1000.0):
    names, dens = parity.default_species(n_spec)
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    modes, nc = parity.make_modes(n_spec, n_class, coarse)
    c = oracle.Cell(n_spec, n_class, 3, int(2.2 * n_part), dens, ve, T, p)
    c.mag[:] = nc * conc_scale / n_part * (1 + 0.5 * np.arange(n_class))
    c.generate(modes, n_part, seed, cell_id, first_id=cell_id << 32)
    c.kmax[:] = oracle.kmax_table(kernel, vs, dens.min(), dens.max(), T, p, const_k, beta)
    return c


@pytest.mark.parametrize("K,n_class", [(8, 1), (3, 1), (4, 2)])
def test_component_lists_follow_the_particles(oracle, K, n_class):
    """particle%component(:) in the oracle (interface/wrf_pmc_driver.F90:1421-1437; the merge rule
    is upstream's, uncited): pinned by what the lists must satisfy whatever the order of events --
    an untruncated list has n_orig(s) entries of source s and spans exactly the particle's least
    and greatest create times; a truncated one is full; a doubled particle carries its original's
    list; and a merged particle's list is its first candidate's followed by its second's (checked
    by replaying one coagulation by hand)."""
    n_part = 1500
    names, dens = parity.default_species(2)
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    modes, nc = parity.make_modes(2, n_class)
    c = oracle.Cell(2, n_class, 3, 4096, dens, ve, n_comp_max=K)
    c.mag[:] = nc * 40.0 / n_part * (1 + 0.5 * np.arange(n_class))
    c.generate(modes, n_part, 9, 0)
    c.lct[:n_part] = c.gct[:n_part] = c.comp_time[:n_part, 0] = 1e-3 * np.arange(n_part)
    assert (c.comp_n[:n_part] == 1).all() and np.array_equal(c.comp_src[:n_part, 0], c.n_orig[:, :n_part].argmax(0))
    c.kmax[:] = oracle.kmax_table(K_BROWN, vs, dens.min(), dens.max(), c.temp, c.pressure)
    next_id = 1 << 40
    for step in range(4):
        before = {int(i): (c.comp_src[k].copy(), c.comp_time[k].copy(), int(c.comp_n[k]))
                  for k, i in enumerate(c.id[:int(c.n_part[0])])}
        st, ns, nco = c.mc_coag(K_BROWN, 900.0, 9, 0, step)
        assert st == 0 and nco > 20
        n = int(c.n_part[0])
        # a particle whose list grew by exactly one other list: the other is a particle that left
        gone = set(before) - set(int(i) for i in c.id[:n])
        seen = 0
        for k in range(n):
            src0, t0, n0 = before[int(c.id[k])]
            n1 = int(c.comp_n[k])
            if n1 == n0 or n1 == K:
                continue
            mine = set(t0[:n0].tolist())
            other = [t for t in c.comp_time[k, :n1].tolist() if t not in mine]
            # its own entries stay together and in order: lists are only ever concatenated
            own_pos = [j for j, t in enumerate(c.comp_time[k, :n1].tolist()) if t in mine]
            assert own_pos == list(range(own_pos[0], own_pos[0] + n0)) and len(other) == n1 - n0
            assert c.comp_time[k, own_pos[0]:own_pos[0] + n0].tolist() == t0[:n0].tolist()
            seen += 1
        assert seen > 10 and (n_class > 1 or len(gone) == nco)
        _, next_id = c.rebalance(np.full(n_class, 0.9 * n_part / n_class), True, True, 9, 0, step, next_id)
    n = int(c.n_part[0])
    tot = c.n_orig[:, :n].sum(0)
    assert np.array_equal(c.comp_n[:n], np.minimum(tot, K)) and (tot > 1).sum() > 100
    whole = tot <= K
    assert whole.any() and (K > 4 or (~whole).any())
    for s in range(3):
        assert np.array_equal((c.comp_src[:n][whole] == s).sum(1), c.n_orig[s, :n][whole])
    t = np.where(c.comp_src[:n] >= 0, c.comp_time[:n], np.nan)
    assert np.array_equal(np.nanmin(t[whole], 1), c.lct[:n][whole])
    assert np.array_equal(np.nanmax(t[whole], 1), c.gct[:n][whole])
    assert (np.nanmin(t[~whole], 1) >= c.lct[:n][~whole]).all() and (np.nanmax(t[~whole], 1) <= c.gct[:n][~whole]).all()
    assert (c.comp_src[:n][np.arange(K)[None, :] >= c.comp_n[:n, None]] == -1).all()
    assert (c.comp_time[:n][np.arange(K)[None, :] >= c.comp_n[:n, None]] == 0).all()
    # doubling copies the list: afterwards slot n + k holds the list of the k-th particle of the class
    src0, t0, wc0 = c.comp_src[:n].copy(), c.comp_time[:n].copy(), c.wclass[:n].copy()
    c.rebalance(np.full(n_class, 3.0 * n), True, False, 9, 0, 99, 1 << 50)
    n2 = int(c.n_part[0])
    assert n2 >= 2 * n
    assert np.array_equal(c.comp_src[:n], src0) and np.array_equal(c.comp_time[:n], t0)
    first = int((wc0 == 0).sum())       # the first doubling: the particles of class 0 in slot order
    assert np.array_equal(c.comp_src[n:n + first], src0[wc0 == 0]) and np.array_equal(c.comp_time[n:n + first], t0[wc0 == 0])
    have = {(tuple(a), tuple(b)) for a, b in zip(src0.tolist(), t0.tolist())}
    assert all((tuple(a), tuple(b)) in have for a, b in zip(c.comp_src[n:n2].tolist(), c.comp_time[n:n2].tolist()))


def test_single_class_conservation(oracle):
    c = _cell(oracle, 3000, 3, 1, K_BROWN, seed=4)
    n0 = 3000
    v0 = c.vol[:, :n0].copy()
    src0 = c.n_orig[:, :n0].sum(1)
    tot = 0
    for step in range(5):
        st, ns, nc = c.mc_coag(K_BROWN, 3000.0, 4, 0, step)
        assert st == 0 and ns >= nc > 0
        tot += nc
    n = int(c.n_part[0])
    assert n == n0 - tot                                    # one particle lost per event
    np.testing.assert_allclose(c.vol[:, :n].sum(1), v0.sum(1), rtol=1e-13)   # species volume
    assert np.array_equal(c.n_orig[:, :n].sum(1), src0)      # constituents
    assert len(np.unique(c.id[:n])) == n                     # ids unique
    # caches are maintained additively by mc_coag (sum of the parents' caches): equal to
    # the species sums up to rounding, and exactly equal again after a refresh
    vt = np.zeros(n); ms = np.zeros(n)
    for s in range(3):
        vt = vt + c.vol[s, :n]; ms = ms + c.density[s] * c.vol[s, :n]
    np.testing.assert_allclose(c.vtot[:n], vt, rtol=1e-14)
    np.testing.assert_allclose(c.mass[:n], ms, rtol=1e-14)
    c.refresh_caches()
    assert np.array_equal(vt, c.vtot[:n]) and np.array_equal(ms, c.mass[:n])
    # the survivor keeps the id of the larger particle: the largest particle's id is original
    assert (c.id[:n] >> 32 == 0).all()


def test_deterministic_and_seed_sensitive(oracle):
    runs = []
    for seed, cid in ((4, 0), (4, 0), (5, 0), (4, 1)):
        c = _cell(oracle, 1500, 2, 1, K_BROWN, seed=4)
        r = [c.mc_coag(K_BROWN, 2000.0, seed, cid, s) for s in range(2)]
        runs.append((r, c.vol[:, :int(c.n_part[0])].copy()))
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])
    assert runs[0][0] != runs[2][0] and runs[0][0] != runs[3][0]


def test_weighted_classes_statistics(oracle):
    """Two weight classes with different magnitudes: physical (number-weighted)
    species volume is conserved in expectation [DeVille et al. 2011]."""
    ratios = []
    for seed in range(24):
        c = _cell(oracle, 1500, 2, 2, K_BROWN, seed=seed, cell_id=seed)
        n0 = 1500
        phys0 = (c.vol[:, :n0] * c.mag[c.wclass[:n0]]).sum()
        for step in range(3):
            c.mc_coag(K_BROWN, 6000.0, seed, seed, step)
        n = int(c.n_part[0])
        ratios.append((c.vol[:, :n] * c.mag[c.wclass[:n]]).sum() / phys0)
    r = np.array(ratios)
    assert abs(r.mean() - 1.0) < 4 * r.std() / math.sqrt(len(r)) + 1e-3


def test_overflow_flags(oracle):
    c = _cell(oracle, 800, 1, 1, K_BROWN, seed=1)
    c.max_events = 5
    st, ns, nc = c.mc_coag(K_BROWN, 30000.0, 1, 0, 0)
    assert st & oracle.ST_EVENT_OVERFLOW and nc == 5
    c = _cell(oracle, 800, 1, 1, K_BROWN, seed=1)
    # appends go behind their list in perm while there is room (16 + n/8 entries);
    # max_appends bounds the ones that do not fit
    c.max_appends = 3
    st, ns, nc = c.mc_coag(K_BROWN, 30000.0, 1, 0, 0)
    assert st == 0 and c.sort_valid[0] == 1
    c = _cell(oracle, 800, 1, 1, K_BROWN, seed=1)
    c.max_appends = 3
    st, ns, nc = c.mc_coag(K_BROWN, 300000.0, 1, 0, 0)
    assert st & oracle.ST_APPEND_OVERFLOW and c.sort_valid[0] == 0


# ----------------------------------------------- analytic Smoluchowski tests

def test_constant_kernel_analytic(oracle):
    """N(t)/N0 = 1/(1 + K N0 t/2) for the constant kernel; ensemble of cells."""
    K = 0.25 / (60 * 2e8)
    n_part, reps = 2000, 12
    t_end, steps = 600.0, 30
    finals, N0 = [], None
    for r in range(reps):
        c = _cell(oracle, n_part, 1, 1, K_CONST, seed=100 + r, cell_id=r, conc_scale=1e-2)
        N0 = c.mag[0] * n_part
        for s in range(steps):
            c.mc_coag(K_CONST, t_end / steps, 100 + r, r, s, const_k=K)
        finals.append(c.mag[0] * c.n_part[0])
    exact = N0 / (1 + K * N0 * t_end / 2)
    m, sd = np.mean(finals), np.std(finals) / math.sqrt(reps)
    assert exact < 0.75 * N0                       # the test exercises real depletion
    assert abs(m - exact) < 4 * sd + 0.01 * exact  # dt-splitting bias is O(1%)


def test_additive_kernel_analytic(oracle):
    """dN/dt = -beta*phi*N for K = beta (v1+v2) (Golovin): N(t) = N0 exp(-beta phi t)."""
    beta = 1000.0
    n_part, reps, steps = 2000, 12, 40
    finals, exact = [], None
    for r in range(reps):
        c = _cell(oracle, n_part, 1, 1, K_ADD, seed=300 + r, cell_id=r, beta=beta)
        N0 = c.mag[0] * n_part
        phi = c.mag[0] * c.vtot[:n_part].sum()
        t_end = 0.5 / (beta * phi)
        for s in range(steps):
            c.mc_coag(K_ADD, t_end / steps, 300 + r, r, s, additive_beta=beta)
        finals.append(c.mag[0] * c.n_part[0] / N0)
        exact = math.exp(-0.5)
    m, sd = np.mean(finals), np.std(finals) / math.sqrt(reps)
    assert abs(m - exact) < 4 * sd + 0.015


# --------------------------------------------------------- rebalance, diags

def test_rebalance_doubling_and_halving(oracle):
    c = _cell(oracle, 1000, 2, 2, K_BROWN, seed=8)
    n_cls0 = np.bincount(c.wclass[:1000], minlength=2)
    mag0 = c.mag.copy()
    physical0 = (c.vol[:, :1000] * c.mag[c.wclass[:1000]]).sum()
    # ideal far above: both classes double (maybe twice) until >= ideal/2
    ideal = n_cls0 * 3.0
    st, nid = c.rebalance(ideal, True, True, 8, 0, 0, next_id=1 << 40)
    n = int(c.n_part[0])
    n_cls = np.bincount(c.wclass[:n], minlength=2)
    assert st == 0 and (n_cls >= ideal / 2).all() and (n_cls == 2 * n_cls0).all()
    assert np.allclose(c.mag, mag0 / 2) and nid == (1 << 40) + n_cls0.sum()
    assert len(np.unique(c.id[:n])) == n
    np.testing.assert_allclose((c.vol[:, :n] * c.mag[c.wclass[:n]]).sum(), physical0, rtol=1e-12)
    # ideal far below: halve until <= 2*ideal; physical volume conserved in expectation
    st, _ = c.rebalance(n_cls0 / 4.0, True, True, 8, 0, 1, next_id=nid)
    n2 = int(c.n_part[0])
    n_cls2 = np.bincount(c.wclass[:n2], minlength=2)
    assert (n_cls2 <= 2 * (n_cls0 / 4.0) + 1).all() and n2 < n
    phys = (c.vol[:, :n2] * c.mag[c.wclass[:n2]]).sum()
    assert abs(phys / physical0 - 1) < 0.45   # few large particles carry the volume: wide spread
    # disabled switches leave the state alone
    before = c.vol.copy()
    c.rebalance(n_cls0 * 100.0, False, False, 8, 0, 2, next_id=0)
    assert np.array_equal(before, c.vol)


def test_diagnostics_against_numpy(oracle):
    c = _cell(oracle, 4000, 3, 2, K_BROWN, seed=9)
    de = np.exp(np.linspace(np.log(1e-9), np.log(1e-3), 101))
    dw = np.log(de[1:]) - np.log(de[:-1])
    i_water = 2
    tn, tm, nd, md = c.diagnostics(i_water, de, dw)
    n = 4000
    dry = c.vol[[0, 1], :n].sum(0)
    diam = 2 * np.cbrt(dry * 0.75 / np.pi)
    mass = (c.density[:, None] * c.vol[:, :n]).sum(0)
    nc = c.mag[c.wclass[:n]]
    assert abs(tn - nc.sum()) <= 1e-12 * tn and abs(tm - (mass * nc).sum()) <= 1e-12 * tm
    b = np.searchsorted(de, diam, side="right") - 1
    ref_n = np.bincount(b, weights=nc, minlength=100) / dw
    ref_m = np.bincount(b, weights=nc * mass, minlength=100) / dw
    np.testing.assert_allclose(nd, ref_n, rtol=1e-11)
    np.testing.assert_allclose(md, ref_m, rtol=1e-11)
    assert abs((nd * dw).sum() - tn) <= 1e-12 * tn   # histogram integrates to the total


def test_generator_statistics(oracle):
    c = _cell(oracle, 20000, 2, 1, K_BROWN, seed=2)
    r = np.cbrt(c.vtot[:20000] * 0.75 / np.pi)
    lr = np.log10(r)
    small = lr < np.log10(2.4e-8)
    # mode 1: gm radius 1e-8, log10 sigma 0.161; number fraction 3.2/(3.2+2.9)
    assert abs(small.mean() - 3.2 / 6.1) < 0.03
    assert abs(np.median(lr[small]) - np.log10(1e-8)) < 0.03


# ------------------------------------------------------------ golden fixtures

def test_golden_kernel_table(oracle):
    g = np.load(os.path.join(GOLDEN, "brownian_kernel_16x16.npz"))
    d = g["diam"]
    for key, (T, p) in (("k_288", (288.15, 101325.0)), ("k_220", (220.0, 25000.0))):
        got = np.array([[oracle.kernel(K_BROWN, math.pi / 6 * a ** 3, math.pi / 6 * a ** 3 * 1770.0,
                                       math.pi / 6 * b ** 3, math.pi / 6 * b ** 3 * 1000.0, T, p)
                         for b in d] for a in d])
        assert np.array_equal(got, g[key])


def test_golden_event_log(oracle):
    """1000-particle single cell, seed 0, 10 steps of 60 s (BASELINE config 1 shape)."""
    with open(os.path.join(GOLDEN, "single_cell_1000_seed0.json")) as f:
        gold = json.load(f)
    from tests.golden.make_golden import run_config1
    got = run_config1(oracle)
    assert got["counters"] == gold["counters"]
    assert got["n_part"] == gold["n_part"]
    assert got["vol_checksum"] == gold["vol_checksum"]
    assert got["id_checksum"] == gold["id_checksum"]


def test_golden_grid_sequence(oracle):
    """Preweight -> transport -> mc_coag -> rebalance on a 3 x 2 x 3 periodic grid with component
    lists (tests/golden/grid_3x2x3_seed5.json): the transport's draw, the exact-count thinning, the
    order of arrivals and the merge order of the lists, pinned across rounds."""
    with open(os.path.join(GOLDEN, "grid_3x2x3_seed5.json")) as f:
        gold = json.load(f)
    from tests.golden.make_golden import run_grid
    got = run_grid(oracle)
    assert got["counters"] == gold["counters"] and got["n_part"] == gold["n_part"]
    for k in ("vol", "id", "wclass", "mag", "comp_source", "comp_time"):
        assert got[k] == gold[k], k
    assert sum(c["moved"] for c in gold["counters"]) > 1000 and gold["counters"][0]["thinned"] > 1000


# ------------------------------------------------ condensation (extension)

def test_condensation_reaches_koehler_equilibrium(oracle):
    dens = np.array([1770.0, 1700.0, 1000.0]); kappa = np.array([0.65, 0.0, 0.0])
    ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    T = 288.15
    c = oracle.Cell(3, 1, 0, 1024, dens, ve, T, 101325.0)
    c.generate([dict(frac=1, gm_radius=5e-8, gsd_log10=0.4, vol_frac=[0.7, 0.3, 0.0])], 1000, 1, 0)
    dry = c.vol[:2, :1000].copy()
    A = 4 * 0.073 * 18e-3 / (8.314472 * T * 1000.0)
    for RH in (0.5, 0.9, 0.3):
        c.condense(kappa, 2, RH, 600.0, 4)
        assert np.array_equal(c.vol[:2, :1000], dry)              # dry species untouched
        vd = dry.sum(0); d3 = (vd + c.vol[2, :1000]) * 6 / np.pi; dd3 = vd * 6 / np.pi
        D = np.cbrt(d3); keff = 0.65 * dry[0] / vd
        s_eq = (d3 - dd3) / (d3 - dd3 * (1 - keff)) * np.exp(A / D)
        np.testing.assert_allclose(s_eq, RH, rtol=2e-6)            # kappa-Koehler equilibrium
    # kinetic limitation: a very short step leaves large particles short of equilibrium
    c.condense(kappa, 2, 0.95, 600.0, 4)
    w_eq = c.vol[2, :1000].copy()
    c.condense(kappa, 2, 0.3, 600.0, 4)
    c.condense(kappa, 2, 0.95, 1e-4, 4)
    big = np.argsort(dry.sum(0))[-20:]
    assert (c.vol[2, big] < 0.9 * w_eq[big]).all() and (c.vol[2, big] > 0).all()
    # kappa = 0 particles take no water
    c2 = oracle.Cell(3, 1, 0, 64, dens, ve, T, 101325.0)
    c2.generate([dict(frac=1, gm_radius=5e-8, gsd_log10=0.2, vol_frac=[0.0, 1.0, 0.0])], 50, 1, 0)
    c2.condense(kappa, 2, 0.9, 600.0, 4)
    assert not c2.vol[2, :50].any()


# ------------------------------------------------------------- transport (f2)

def _reference_sequential_sample(rng, n_from, prob, ratio):
    """Transcription of the pick loop of aero_state_sample_particles_by_class
    (interface/wrf_pmc_trans_aero.F90:2236-2300), counting adds and removes:
    n_transfer is Binomial(n, p) when ratio >= 1, Poisson(n p / ratio) otherwise (:2236-2241);
    each pick is uniform over the current list (:2249-2252); equal mixing ratios add and
    remove, a smaller destination weight always adds and removes with probability
    to/from, a larger one always removes and adds with probability from/to (:2254-2273)."""
    n_transfer = rng.binomial(n_from, prob) if ratio >= 1.0 else rng.poisson(n_from * prob / ratio)
    n_list, added, removed = n_from, 0, 0
    for _ in range(n_transfer):
        if n_list <= 0:
            break
        if ratio == 1.0:
            add, rem = True, True
        elif ratio < 1.0:
            add, rem = True, rng.uniform() < ratio
        else:
            add, rem = rng.uniform() < 1.0 / ratio, True
        added += add
        if rem:
            removed += 1
            n_list -= 1
    return added, removed


@pytest.mark.parametrize("ratio", [1.0, 2.5, 0.4])
def test_transport_matches_reference_sampling_law(oracle, ratio):
    """The per-particle form used here against a transcription of the reference's sequential
    sampler: same mean number of removals and arrivals (they are equal in distribution for
    ratio >= 1; for ratio < 1 the means agree and the reference's list shrinkage is O(p))."""
    n_part, p, trials = 4000, 0.05, 40
    dens = np.array([1500.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    rng = np.random.RandomState(7)
    ref = np.array([_reference_sequential_sample(rng, n_part, p, ratio) for _ in range(400)])
    got = []
    for t in range(trials):
        cells = []
        for c in range(2):                      # two cells side by side, closed box: 0 <-> 1
            oc = oracle.Cell(1, 1, 0, 3 * n_part, dens, ve)
            oc.mag[:] = 1e6 if c == 0 else 1e6 * ratio   # equal volumes: ratio = mag_to / mag_from
            oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0])], n_part, t, c,
                        first_id=c << 32)
            cells.append(oc)
        ph = np.zeros((2, 1, 4)); ph[0, 0, 1] = p   # only cell 0 sends, eastwards
        next_id = np.array([1 << 40, 2 << 40], dtype=np.int64)
        st, nm, na = oracle.transport(cells, 2, 1, 1, ph, np.zeros((2, 1, 1)), np.ones(2), t, 0, 0,
                                      next_id, False, False)
        assert st == 0 and nm[1] == 0 and na[0] == 0
        assert int(cells[0].n_part[0]) == n_part - nm[0] and int(cells[1].n_part[0]) == n_part + na[1]
        ids = np.concatenate([cells[0].id[:int(cells[0].n_part[0])], cells[1].id[:int(cells[1].n_part[0])]])
        assert len(np.unique(ids)) == len(ids)
        got.append((na[1], nm[0]))
    got = np.array(got)
    for col, name in ((0, "added"), (1, "removed")):
        se = math.sqrt(ref[:, col].var() / len(ref) + got[:, col].var() / trials)
        assert abs(ref[:, col].mean() - got[:, col].mean()) < 4 * se + 0.02 * ref[:, col].mean(), name
    assert abs(got[:, 1].mean() - n_part * p) < 4 * math.sqrt(n_part * p / trials)   # removals = n p
    assert abs(got[:, 0].mean() - n_part * p / ratio) < 0.06 * n_part * p / ratio     # arrivals = n p / r


def _dep_velocity_numpy(d, rho, temp, rrho, ustar, ra, alpha, big_a):
    """Independent transcription of wrf_pmc_dep_aero.F90:265-420 (updated_version branch) in
    numpy float64, with the reference's default-real literals as float32 values."""
    f32 = np.float32
    lam = 0.065e-6
    mu, g, kb = 1.78e-5, 9.80665, 1.3806505e-23
    cc = 1.0 + (2.0 * lam / d) * (float(f32(1.257)) + float(f32(.4)) * np.exp(-(float(f32(1.1)) * d) / (2.0 * lam)))
    vs = d ** 2 * rho * g * cc / (18.0 * mu)
    diff = kb * temp * cc / (3.0 * np.pi * mu * d)
    sc = 1.45e-5 / diff
    st = vs * ustar ** 2 / (mu * rrho) if big_a > 1.0 else vs * ustar / (g * big_a)
    eb = float(f32(.2)) * sc ** (-float(f32(2.0) / f32(3.0)))
    eim = float(f32(0.4)) * (st / (alpha + st)) ** 2.0
    ein = 0.5 * (d / big_a) ** float(f32(0.8))
    r1 = np.exp(-np.sqrt(st))
    rs = 1.0 / (3.0 * ustar * (eb + eim + ein) * r1)
    return 1.0 / (ra + rs + ra * rs * vs) + vs


@pytest.mark.parametrize("nx,ny", [(1, 3), (3, 1), (1, 1)])
def test_a_periodic_direction_one_cell_wide_is_closed(oracle, nx, ny):
    """With one cell across a periodic direction the cell would be its own neighbour: nothing moves
    that way (pmcx.h, pmcx_transport) -- in the transport and in the expectation the preweight
    forms -- and nobody is lost: the outcome is the one with those probabilities set to zero."""
    nz, n_part = 2, 300
    dens = np.array([1800.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    n = nx * nz * ny
    rng = np.random.RandomState(4)
    ph = rng.uniform(0.05, 0.2, (n, 1, 4)); pv = rng.uniform(0.0, 0.1, (n, 1, nz)); cv = rng.uniform(0.8, 1.2, n)
    ph0 = ph.copy()
    if nx == 1:
        ph0[:, :, 0:2] = 0.0
    if ny == 1:
        ph0[:, :, 2:4] = 0.0
    outs = []
    for probs in (ph, ph0):
        cells = []
        for c in range(n):
            oc = oracle.Cell(1, 1, 0, 4 * n_part, dens, ve)
            oc.mag[:] = 1e6 * (1 + c)
            oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0])], n_part, 7, c, first_id=c << 32)
            cells.append(oc)
        next_id = (np.arange(n, dtype=np.int64) << 32) + (1 << 24)
        ex, nr = oracle.transport_preweight(cells, nx, nz, ny, probs, pv, cv, np.full((n, 1), float(n_part)), 7, 0, 0)
        st, nm, na = oracle.transport(cells, nx, nz, ny, probs, pv, cv, 7, 0, 0, next_id)
        assert st == 0
        outs.append((ex, nm, na, [oc.id[:int(oc.n_part[0])].copy() for oc in cells]))
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    if nx == 1 and ny == 1:     # only the levels exchange
        assert a[1].sum() > 0 and a[1].sum() < 0.2 * n * n_part


def test_weight_preparation_of_an_add(oracle):
    """aero_state_prepare_weight_for_add [upstream, uncited] in front of every mode's Poisson sample
    (the reference hands allow_doubling / allow_halving to every add, wrf_pmc_driver.F90:207-210,
    1920-1922, wrf_pmc_trans_aero.F90:1030-1032): when the class would leave [ideal / 2, 2 ideal]
    its weight is scaled so that it comes to the ideal count -- an exact-count thinning above, copies
    below -- and the number concentration stays what it was in expectation.  Off by default."""
    dens = np.array([1800.0, 1000.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    ideal, mag0 = 400.0, 1e6
    mode = dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[0.5, 0.5], wclass=0, source=1)

    def cell(n, prep=True, flags=(True, True), seed=0):
        oc = oracle.Cell(2, 1, 2, 4096, dens, ve, n_comp_max=3)
        oc.mag[:] = mag0
        oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0, 0.0], source=0)], n, seed, 0, first_id=1 << 40)
        if prep:
            oc.set_weight_preparation([ideal], *flags)
        return oc

    stats = {"over": [], "under": [], "empty": []}
    for t in range(40):
        # 1.9 ideal present, 0.5 ideal on their way: above 2 ideal -> thinned to the ideal count
        oc = cell(760, seed=t)
        em = [dict(mode, num_conc=0.5 * ideal * mag0 / 10.0)]           # rate: mean = rate dt / magnitude
        st, nid, ne, ni, no = oc.scenario_update(10.0, 10.0, t, 0, 3, em, 0.0, None, 1)
        n = int(oc.n_part[0])
        assert st == 0 and oc.mag[0] == mag0 * 2.4                      # ratio = (760 + 200) / 400
        kept = n - ne
        assert kept in (int(760 / 2.4), int(760 / 2.4) + 1)             # prob_round(n (1 - 1 / ratio)) left
        assert len(np.unique(oc.id[:n])) == n and nid == 1 + ne
        stats["over"].append((n, ne))
        # 0.2 ideal present, 0.1 ideal on their way: below ideal / 2 -> every particle stands for 1 / 0.3
        oc = cell(80, seed=100 + t)
        before = oc.vtot[:80].copy()
        em = [dict(mode, num_conc=0.1 * ideal * mag0 / 10.0)]
        st, nid, ne, ni, no = oc.scenario_update(10.0, 10.0, t, 0, 3, em, 0.0, None, 1)
        n = int(oc.n_part[0])
        assert st == 0 and oc.mag[0] == mag0 * (120.0 / 400.0)
        copies = n - ne - 80
        assert 2 * 80 <= copies <= 3 * 80                               # 3 or 4 of each: 2 or 3 copies
        assert np.array_equal(oc.vtot[:80], before)                     # the originals stay where they are
        assert np.isin(oc.vtot[80:80 + copies], before).all()           # the copies behind them, then the new ones
        assert (oc.n_orig[1, 80 + copies:n] == 1).all() and (oc.comp_n[80:80 + copies] == 1).all()
        assert len(np.unique(oc.id[:n])) == n and nid == 1 + copies + ne
        stats["under"].append((n, ne))
        # an emptied boundary cell at a stale weight: the refill comes at the weight that gives the ideal count
        oc = cell(300, seed=200 + t)
        oc.mag[:] = mag0 / 500.0                                        # (a copy of a much-doubled neighbour)
        bg = [dict(mode, num_conc=2.0 * ideal * mag0)]                  # mixing ratio, density 1
        st, nid, na, nr = oc.boundary_resample(True, bg, 1.0, 0.0, t, 0, 3, 1)
        assert st == 0 and nr == 300 and oc.mag[0] == 2.0 * ideal * mag0 / ideal
        stats["empty"].append((int(oc.n_part[0]), na))
    for k, v in stats.items():
        v = np.array(v, dtype=float)
        assert abs(v[:, 0].mean() - ideal) < 4 * math.sqrt(ideal / len(v)) + 1, (k, v[:, 0].mean())
    # flags: with particles present a scaling needs its flag; off by default
    oc = cell(760, flags=(True, False))
    st, nid, ne, *_ = oc.scenario_update(10.0, 10.0, 1, 0, 3, [dict(mode, num_conc=0.5 * ideal * mag0 / 10.0)], 0.0, None, 1)
    assert oc.mag[0] == mag0 and int(oc.n_part[0]) == 760 + ne
    oc = cell(80, flags=(False, True))
    st, nid, ne, *_ = oc.scenario_update(10.0, 10.0, 1, 0, 3, [dict(mode, num_conc=0.1 * ideal * mag0 / 10.0)], 0.0, None, 1)
    assert oc.mag[0] == mag0 and int(oc.n_part[0]) == 80 + ne
    # the characteristic factor: the weight is prepared for the particles expected over the sampling
    # timescale, max(1, min(3600 s, elapsed_time)) / delta_t times the call's mean (driver:1916-1918)
    oc = cell(760)
    st, nid, ne, *_ = oc.scenario_update(10.0, 7200.0, 1, 0, 3, [dict(mode, num_conc=0.5 * ideal * mag0 / 10.0)], 0.0, None, 1)
    assert oc.mag[0] == mag0 * ((760 + 200 * 360.0) / ideal) and int(oc.n_part[0]) - ne in (4, 5)
    oc = cell(760)
    st, nid, ne, *_ = oc.scenario_update(10.0, 0.0, 1, 0, 3, [dict(mode, num_conc=0.5 * ideal * mag0 / 10.0)], 0.0, None, 1)
    assert oc.mag[0] == mag0 and int(oc.n_part[0]) == 760 + ne          # 760 + 200 / 10 stays inside 2 ideal
    oc = cell(760, prep=False)
    st, nid, ne, *_ = oc.scenario_update(10.0, 10.0, 1, 0, 3, [dict(mode, num_conc=0.5 * ideal * mag0 / 10.0)], 0.0, None, 1)
    assert oc.mag[0] == mag0 and int(oc.n_part[0]) == 760 + ne


@pytest.mark.parametrize("ratio", [0.4, 0.1])
def test_prototype_of_the_reference_transport_law_for_small_ratios(oracle, ratio):
    """Law 1 of the transport (the default): the r < 1 branch by Poisson thinning
    of the reference's picks -- movers arrive once, particles that stay emit the other copies.
    Against the transcription of the reference's pick loop: the same mean of arrivals and removals
    AND the same variance of the arrivals (n p / r), where the library's law (0) has
    n p (2 - r - p) / r^2."""
    n_part, p, trials = 2000, 0.04, 150
    dens = np.array([1500.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    rng = np.random.RandomState(11)
    ref = np.array([_reference_sequential_sample(rng, n_part, p, ratio) for _ in range(1500)])
    got = {0: [], 1: []}
    try:
        for law in (0, 1):
            oracle.set_transport_law(law)
            for t in range(trials):
                cells = []
                for c in range(2):
                    oc = oracle.Cell(1, 1, 0, int(n_part * (1 + 3 * p / ratio)) + n_part, dens, ve)
                    oc.mag[:] = 1e6 if c == 0 else 1e6 * ratio
                    oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0])], n_part, t, c, first_id=c << 32)
                    cells.append(oc)
                ph = np.zeros((2, 1, 4)); ph[0, 0, 1] = p
                next_id = np.array([1 << 40, 2 << 40], dtype=np.int64)
                st, nm, na = oracle.transport(cells, 2, 1, 1, ph, np.zeros((2, 1, 1)), np.ones(2), 1000 + t, 0, 0, next_id,
                                              False, False)
                assert st == 0 and int(cells[0].n_part[0]) == n_part - nm[0] and int(cells[1].n_part[0]) == n_part + na[1]
                ids = np.concatenate([oc.id[:int(oc.n_part[0])] for oc in cells])
                assert len(np.unique(ids)) == len(ids)
                got[law].append((na[1], nm[0]))
    finally:
        oracle.set_transport_law()
    g0, g1 = np.array(got[0], float), np.array(got[1], float)
    mean_add = n_part * p / ratio
    for g in (g0, g1):                                         # both laws: the reference's means
        assert abs(g[:, 0].mean() - ref[:, 0].mean()) < 4 * math.sqrt(g[:, 0].var() / trials + ref[:, 0].var() / len(ref)) + 0.02 * mean_add
        assert abs(g[:, 1].mean() - n_part * p) < 4 * math.sqrt(n_part * p / trials) + 0.03 * n_part * p
    v_ref, v0, v1 = ref[:, 0].var(ddof=1), g0[:, 0].var(ddof=1), g1[:, 0].var(ddof=1)
    assert abs(v_ref / (n_part * p / ratio) - 1) < 0.15                         # the reference: Poisson picks
    assert abs(v1 / v_ref - 1) < 0.35, (v1, v_ref)                             # law 1: the reference's variance
    assert v0 > (1.5 if ratio == 0.4 else 8.0) * v_ref, (v0, v_ref)           # law 0: (2 - r - p) / r times it


@pytest.mark.parametrize("p,ratio", [(0.4, 0.25), (0.3, 0.1), (0.6, 0.03)])
def test_transport_law_1_keeps_the_mean_where_movers_and_emitters_overlap(oracle, p, ratio):
    """Law 1 where sum p + 1 - exp(-sum mu) > 1 (a large probability to leave and a small weight
    ratio; from r of about 0.4 at p = 0.5): the emitters' interval lies at the top of the unit interval
    and overlaps the movers' -- a particle in the overlap sends its copies and moves.  Arrivals keep
    the mean n p / r of the reference's Poisson(n p / r) picks (wrf_pmc_trans_aero.F90:2236-2241; until
    round 4 the interval was cut at 1 and the mean fell short: 0.89 of it at (0.4, 0.25), 0.77 at
    (0.3, 0.1)), removals the mean n p; against the transcription of the reference's pick loop; no
    status bit, every id once."""
    n_part, trials = 1000, 30
    dens = np.array([1500.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    rng = np.random.RandomState(3)
    ref = np.array([_reference_sequential_sample(rng, n_part, p, ratio) for _ in range(200)])
    got = []
    for t in range(trials):
        cells = []
        for c in range(2):
            oc = oracle.Cell(1, 1, 0, int(n_part * (1 + 1.3 * p / ratio)) + n_part, dens, ve)
            oc.mag[:] = 1e6 if c == 0 else 1e6 * ratio
            oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0])], n_part, t, c, first_id=c << 32)
            cells.append(oc)
        ph = np.zeros((2, 1, 4)); ph[0, 0, 1] = p
        next_id = np.array([1 << 40, 2 << 40], dtype=np.int64)
        st, nm, na = oracle.transport(cells, 2, 1, 1, ph, np.zeros((2, 1, 1)), np.ones(2), 500 + t, 0, 0, next_id,
                                      False, False)
        assert st == 0 and int(cells[0].n_part[0]) == n_part - nm[0] and int(cells[1].n_part[0]) == n_part + na[1]
        ids = np.concatenate([oc.id[:int(oc.n_part[0])] for oc in cells])
        assert len(np.unique(ids)) == len(ids)
        got.append((na[1], nm[0]))
    got = np.array(got, float)
    mean_add = n_part * p / ratio
    assert abs(got[:, 0].mean() / mean_add - 1) < 4 * math.sqrt(1.0 / (mean_add * trials)) + 0.005, got[:, 0].mean() / mean_add
    assert abs(got[:, 0].mean() - ref[:, 0].mean()) < 0.01 * mean_add
    assert abs(got[:, 1].mean() - n_part * p) < 4 * math.sqrt(n_part * p * (1 - p) / trials) + 0.01 * n_part * p
    # the variance stays of the reference's order (Poisson picks: n p / r), nowhere near law 0's
    assert got[:, 0].var(ddof=1) < 2.0 * mean_add


def test_transport_emitter_count_at_large_rates(oracle):
    """The copy count of an emitter for sum mu from small to beyond what exp(-sum mu) can represent
    (the zero-truncated walk's first term underflows near 745: the plain Poisson count takes over
    above 32, where the truncation is below 2^-46): arrivals keep the mean n p / r at r = 1e-3, and a
    ratio that asks for more than the guard of 4096 copies per emitter raises its own status bit
    instead of running every emitter to the guard."""
    dens = np.array([1500.0]); ve, _ = oracle.make_bin_tables(80, 1e-10, 1e-2)
    for ratio, p, n_part, want_bit in ((1e-3, 0.5, 40, 0), (5e-5, 0.5, 8, 128)):
        tot, exp = 0.0, 0.0
        bits = 0
        for t in range(6):
            cells = []
            for c in range(2):
                oc = oracle.Cell(1, 1, 0, 60000, dens, ve)
                oc.mag[:] = 1e6 if c == 0 else 1e6 * ratio
                oc.generate([dict(frac=1, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1.0])], n_part, t, c, first_id=c << 32)
                cells.append(oc)
            ph = np.zeros((2, 1, 4)); ph[0, 0, 1] = p
            next_id = np.array([1 << 40, 2 << 40], dtype=np.int64)
            st, nm, na = oracle.transport(cells, 2, 1, 1, ph, np.zeros((2, 1, 1)), np.ones(2), 900 + t, 0, 0, next_id,
                                          False, False)
            bits |= st
            tot += na[1]; exp += n_part * p / ratio
        assert bits == want_bit, bits
        if not want_bit:
            assert abs(tot / exp - 1) < 4 * math.sqrt(1.0 / exp) + 0.01, tot / exp
        else:
            assert tot <= 6 * n_part * (4096 + 1)


def test_dry_deposition_velocity(oracle):
    """Zhang (2001) deposition velocity (wrf_pmc_dep_aero.F90:265-420): equal to an independent
    numpy transcription to rounding, U-shaped in diameter, settling-dominated for coarse
    particles, bounded by 1/ra + vs."""
    rng = np.random.RandomState(3)
    for big_a, alpha in ((0.002, 1.2), (0.010, 1.3), (9.999, 100.0)):
        for _ in range(200):
            d = 10 ** rng.uniform(-8.5, -4.5); rho = rng.uniform(900, 2600)
            T = rng.uniform(230, 305); rrho = 1.0 / rng.uniform(0.9, 1.3)
            us = rng.uniform(0.05, 0.8); ra = rng.uniform(5, 300)
            got = oracle.dep_velocity(d, rho, T, rrho, us, ra, alpha, big_a)
            ref = _dep_velocity_numpy(d, rho, T, rrho, us, ra, alpha, big_a)
            assert abs(got - ref) <= 1e-12 * ref
            vs = d ** 2 * rho * 9.80665 / (18 * 1.78e-5)
            assert vs < got < 1.0 / ra + vs * (1 + 0.3) + 1e-9 or d < 3e-7
    v = [oracle.dep_velocity(d, 1800.0, 288.0, 1 / 1.2, 0.3, 50.0, 1.2, 0.002) for d in (1e-8, 3e-7, 1e-5)]
    assert v[0] > 5 * v[1] and v[2] > 5 * v[1]          # minimum in the accumulation mode
    assert 1e-4 < v[1] < 2e-3 and 2e-3 < v[0] < 2e-2    # m/s, the textbook range


def test_dry_deposition_removal_statistics(oracle):
    """dry_dep_aero_state (wrf_pmc_dep_aero.F90:154-272): every particle goes with probability
    (dt/dz) vd; the number removed matches the sum of the probabilities, survivors keep their
    order, the sort is invalidated."""
    dens = np.array([1800.0])
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    modes = [dict(frac=1.0, gm_radius=2e-8, gsd_log10=0.4, vol_frac=[1.0], wclass=0, source=0)]
    c = oracle.Cell(1, 1, 1, 8192, dens, ve, 285.0, 1e5)
    c.mag[:] = 1e6
    c.generate(modes, 8000, 5, 0)
    v0, ids0 = c.vtot[:8000].copy(), c.id[:8000].copy()
    dt, dz, ra, us = 60.0, 50.0, 40.0, 0.4
    diam = 2 * np.cbrt(v0 * 0.75 / np.pi)
    prob = np.array([dt / dz * oracle.dep_velocity(d, 1800.0, 285.0, 1 / 1.2, us, ra, 1.2, 0.002) for d in diam])
    nd, bad = c.dry_deposition(dt, 5, 0, 0, ra, us, 1.2, 0.002, dz, 1 / 1.2)
    assert bad == 0 and int(c.n_part[0]) == 8000 - nd
    assert abs(nd - prob.sum()) < 4 * np.sqrt((prob * (1 - prob)).sum())
    keep = np.isin(ids0, c.id[:8000 - nd])
    assert np.array_equal(ids0[keep], c.id[:8000 - nd])      # stable
    assert np.array_equal(v0[keep], c.vtot[:8000 - nd])
    assert int(c.sort_valid[0]) == 0
    nd2, bad2 = c.dry_deposition(1e6, 5, 0, 1, ra, us, 1.2, 0.002, dz, 1 / 1.2)   # absurd step
    assert bad2 == 1 and int(c.n_part[0]) == 0


def test_aerodynamic_resistance_and_landuse_tables():
    """Host-side inputs of dry deposition: depvel_partmc (wrf_pmc_dep.F90:188-213) and the
    Zhang land-use selection (wrf_pmc_dep_aero.F90:66-151)."""
    from wrf_partmc_amd.data import aerodynamic_resistance, zhang_surface_parameters
    ra = aerodynamic_resistance([0.0, 0.02, -0.02, 1e-7], 0.1, 0.3)
    assert abs(ra[0] - 0.74 * np.log(2.0 / 0.1) / (0.4 * 0.3)) < 1e-12
    assert ra[3] == ra[0]                       # |rmol| < 1e-6 is neutral
    assert ra[1] > ra[0] > ra[2] > 0            # stable > neutral > unstable
    assert aerodynamic_resistance(0.0, 0.1, 0.0) == 0.74 * np.log(20.0) / (0.4 * 1e-4)
    alpha, big_a = zhang_surface_parameters([1, 7, 16, 24, 0, 99])
    w = lambda xs: [float(np.float32(x)) for x in xs]   # (default-real literals of the reference, widened)
    assert list(alpha) == w([1.5, 1.2, 100.0, 50.0, 1.2, 1.2])     # urban, grass, water, ice, fallback
    assert list(big_a) == w([0.010, 0.002, 9.999, 9.999, 0.002, 0.002])


def test_cdf_codec_against_scipy(tmp_path):
    """The self-contained NetCDF encoder/decoder of ncio.py (no netCDF library in this image):
    its CDF-2 files are read by scipy.io.netcdf_file, it reads scipy's, and the CDF-5 variant
    (needed for the reference's 64-bit aero_id / next_id) differs only in 8-byte counts."""
    from scipy.io import netcdf_file
    from wrf_partmc_amd import ncio
    rng = np.random.RandomState(0)
    a = rng.standard_normal((3, 5)); b = rng.randint(-9, 9, 7).astype(np.int32)
    w = ncio.CdfWriter(version=2)
    w.att("title", "t"); w.att("n", np.int32(4))
    w.dim("x", 3); w.dim("y", 5); w.dim("z", 7)
    w.var("a", ["x", "y"], a, unit="m", long_name="an array")
    w.var("b", ["z"], b)
    p = str(tmp_path / "mine.nc"); w.write(p)
    f = netcdf_file(p, "r", mmap=False)
    assert f.dimensions == {"x": 3, "y": 5, "z": 7} and f.title == b"t" and f.n == 4
    assert np.array_equal(f.variables["a"][:], a) and f.variables["a"].unit == b"m"
    assert f.variables["a"].dimensions == ("x", "y") and np.array_equal(f.variables["b"][:], b)
    f.close()
    q = str(tmp_path / "theirs.nc")
    g = netcdf_file(q, "w", version=2)
    g.history = "h"; g.createDimension("p", 4); g.createDimension("s", 2)
    v = g.createVariable("m", "d", ("s", "p")); v[:] = np.arange(8.0).reshape(2, 4); v.unit = "kg"
    g.createVariable("k", "i", ("p",))[:] = [1, 2, 3, 4]
    g.close()
    dims, gatts, vs = ncio.read_cdf(q)
    assert dims == {"p": 4, "s": 2} and gatts["history"] == "h"
    assert np.array_equal(vs["m"][0], np.arange(8.0).reshape(2, 4)) and vs["m"][1]["unit"] == "kg"
    assert vs["m"][2] == ["s", "p"] and list(vs["k"][0]) == [1, 2, 3, 4]
    w5 = ncio.CdfWriter(version=5)
    w5.dim("n", 3)
    big = np.array([1 << 40, -(1 << 50), 7], dtype=np.int64)
    w5.var("id", ["n"], big, nc_type=ncio.NC_INT64, long_name="ids")
    p5 = str(tmp_path / "five.nc"); w5.write(p5)
    raw = open(p5, "rb").read()
    assert raw[:4] == b"CDF\x05" and raw[4:12] == b"\x00" * 8          # magic, numrecs (8 bytes)
    dims, _, vs = ncio.read_cdf(p5)
    assert dims == {"n": 3} and np.array_equal(vs["id"][0], big) and vs["id"][0].dtype == np.int64


def test_cdf5_encoder_against_a_hand_assembled_file(tmp_path):
    """CDF-5 is never checked against a netCDF library here (none in the image): instead the
    encoder must reproduce, byte for byte, a file assembled by hand from the format grammar
    (tests/golden/cdf5_minimal.hex, every field annotated), and the decoder must read it."""
    from wrf_partmc_amd import ncio
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "golden", "cdf5_minimal.hex")).read()
    want = bytes.fromhex("".join(ln.split("#")[0] for ln in text.splitlines()).replace(" ", ""))
    assert len(want) == 300
    w = ncio.CdfWriter(version=5)
    w.att("title", "t")
    w.dim("n", 3)
    w.var("aero_id", ["n"], np.array([1, 1 << 40, -5], dtype=np.int64), unit="1")
    w.var("x", ["n"], np.array([0.5, -1.0, 2.0]))
    assert w.tobytes() == want
    p = str(tmp_path / "hand.nc")
    open(p, "wb").write(want)
    dims, gatts, vs = ncio.read_cdf(p)
    assert dims == {"n": 3} and gatts == {"title": "t"}
    assert vs["aero_id"][0].dtype == np.int64 and list(vs["aero_id"][0]) == [1, 1 << 40, -5]
    assert vs["aero_id"][1] == {"unit": "1"} and list(vs["x"][0]) == [0.5, -1.0, 2.0]


def test_column_writer_against_the_assembled_cdf5_column(tmp_path):
    """SURVEY f3 / the whole variable set of output_column_to_file_new (wrf_pmc_driver.F90:1372-1379,
    1478-1653), aero_component_* with per-constituent create times included: write_column must
    reproduce, byte for byte, the file tests/golden/make_cdf5_column.py assembles from the format
    grammar with code of its own (tests/golden/cdf5_column.hex, every field annotated), and
    read_column must give back the values that script states."""
    from wrf_partmc_amd import ncio
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "golden", "cdf5_column.hex")).read()
    want = bytes.fromhex("".join(ln.split("#")[0] for ln in text.splitlines()).replace(" ", ""))
    assert len(want) == 5988 and want[:4] == b"CDF\x05"
    dens = np.array([1800.0, 1700.0])
    vol = np.array([[1e-21, 2.5e-21, 4e-22], [0.0, 5e-22, 6e-22]])
    ids = np.array([7, (3 << 32) + 11, (1 << 40) + 2], dtype=np.int64)
    lct, gct = np.array([0.0, 12.5, 300.0]), np.array([0.0, 86400.25, 300.0])
    n_orig = np.array([[1, 2, 0], [0, 1, 1]])
    comp_len = np.array([1, 3, 1]); comp_source = np.array([[0, -1, -1], [1, 0, 0], [1, -1, -1]])
    comp_time = np.array([[0.0, 0, 0], [86400.25, 12.5, 3600.0], [300.0, 0, 0]])
    path = str(tmp_path / "col.nc")
    ncio.write_column(path, species_names=["SO4", "BC"], density=dens, part_offset=[0, 2, 3], vol=vol,
                      wclass=[0, 0, 0], ids=ids, least_ct=lct, greatest_ct=gct, n_orig=n_orig,
                      weight_magnitude=[[2.5e7], [1.25e7]], next_id=(1 << 40) + 3, time=7200.0, del_t=60.0,
                      index=120, kappa=[0.65, 0.0], molec_weight=[0.096, 0.001], source_names=["traffic", "sea"],
                      gas_names=["o3"], gas_mixing_ratio=[[41.5], [38.25]],
                      uuid="c0ffee00-0000-4000-8000-000000000001", cell_ix=12, cell_iy=7,
                      comp_len=comp_len, comp_source=comp_source, comp_time=comp_time)
    got = open(path, "rb").read()
    if got != want:
        k = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
        raise AssertionError(f"first difference at byte {k}: {got[k - 8:k + 24].hex()} != {want[k - 8:k + 24].hex()}")
    fx = str(tmp_path / "fixture.nc")
    open(fx, "wb").write(want)
    col = ncio.read_column(fx)
    assert list(col["part_offset"]) == [0, 2, 3] and col["next_id"] == (1 << 40) + 3 and col["index"] == 120
    assert np.array_equal(col["ids"], ids) and np.array_equal(col["n_orig"], n_orig)
    assert np.array_equal(col["comp_len"], comp_len) and np.array_equal(col["comp_source"], comp_source)
    assert np.array_equal(col["comp_time"], comp_time)
    assert np.array_equal(col["least_ct"], lct) and np.array_equal(col["greatest_ct"], gct)
    np.testing.assert_allclose(col["vol"], vol, rtol=3e-16, atol=0)
    assert np.array_equal(col["weight_magnitude"], [[2.5e7], [1.25e7]]) and np.array_equal(col["gas_mixing_ratio"], [[41.5], [38.25]])
    # without the lists the writer expands the histogram: the same particles, source order, bounds as times
    ncio.write_column(path, species_names=["SO4", "BC"], density=dens, part_offset=[0, 2, 3], vol=vol,
                      wclass=[0, 0, 0], ids=ids, least_ct=lct, greatest_ct=gct, n_orig=n_orig,
                      weight_magnitude=[[2.5e7], [1.25e7]], next_id=1)
    col = ncio.read_column(path)
    assert col["comp_source"].tolist() == [[0, -1, -1], [0, 0, 1], [1, -1, -1]]
    assert col["comp_time"].tolist() == [[0.0, 0, 0], [12.5, 86400.25, 86400.25], [300.0, 0, 0]]
    # a netCDF-4 (HDF5) container is named for what it is
    open(path, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(ValueError, match="nccopy -k cdf5"):
        ncio.read_cdf(path)


def test_column_file_round_trip(oracle, tmp_path):
    """a column written in the layout of output_column_to_file_new
    (wrf_pmc_driver.F90:1508-1653) and read back the way init_read_in_restart does
    (wrf_pmc_init.F90:1084-1286): names, dimensions, 1-based indices, mass = vol * density."""
    from wrf_partmc_amd import ncio
    from wrf_partmc_amd.data import aero_data_arrays
    dens, kappa, mw, names = aero_data_arrays()
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    vf = np.zeros(20); vf[[0, 3, 17]] = [0.36, 0.14, 0.5]
    modes = [dict(frac=0.6, gm_radius=1e-8, gsd_log10=0.161, vol_frac=vf, wclass=0, source=0),
             dict(frac=0.4, gm_radius=5.8e-8, gsd_log10=0.217, vol_frac=vf, wclass=1, source=2)]
    counts = [300, 0, 120, 77]
    cells = []
    for k, n in enumerate(counts):
        c = oracle.Cell(20, 2, 3, 512, dens, ve, 280.0, 9e4, n_comp_max=2)
        c.mag[:] = [1e7 + k, 3e7 + k]
        c.generate(modes, n, 2, k, first_id=(k << 40) + 5)
        c.lct[:n] = c.gct[:n] = c.comp_time[:n, 0] = 10.0 * k + 0.01 * np.arange(n)
        c.kmax[:] = oracle.kmax_table(4, vs, dens.min(), dens.max(), 280.0, 9e4)
        for s in range(3):
            c.mc_coag(4, 4000.0, 2, k, s)           # merged particles: several constituents
        cells.append(c)
    n_live = [int(c.n_part[0]) for c in cells]
    po = np.concatenate([[0], np.cumsum(n_live)])
    cat = lambda f: np.concatenate([f(c, n) for c, n in zip(cells, n_live)], axis=-1)
    state = dict(vol=cat(lambda c, n: c.vol[:, :n]), wclass=cat(lambda c, n: c.wclass[:n]),
                 ids=cat(lambda c, n: c.id[:n]), least_ct=cat(lambda c, n: c.lct[:n]),
                 greatest_ct=cat(lambda c, n: c.gct[:n]), n_orig=cat(lambda c, n: c.n_orig[:, :n]))
    lists = dict(comp_len=cat(lambda c, n: c.comp_n[:n]), comp_source=cat(lambda c, n: c.comp_src[:n].T).T,
                 comp_time=cat(lambda c, n: c.comp_time[:n].T).T)
    wmag = np.array([c.mag for c in cells])
    path = ncio.column_filename(str(tmp_path / "out"), 12, 7, 42)
    assert path.endswith("out_012_007_00000042.nc")
    ncio.write_column(path, species_names=names, density=dens, part_offset=po, weight_magnitude=wmag,
                      next_id=(9 << 40) + 1, time=126.0, del_t=3.0, index=42, kappa=kappa,
                      molec_weight=mw, gas_names=["so2", "o3"], gas_mixing_ratio=np.ones((4, 2)), **state, **lists)
    dims, gatts, v = ncio.read_cdf(path)
    assert dims["number_levels"] == 4 and dims["aero_particle"] == po[-1] and dims["aero_species"] == 20
    assert dims["aero_weight_class"] == 2 and dims["aero_weight_group"] == 1
    assert v["aero_particle_mass"][2] == ["aero_species", "aero_particle"] and v["aero_particle_mass"][1]["unit"] == "kg"
    assert v["weight_magnitude"][2] == ["aero_weight_class", "aero_weight_group", "number_levels"]
    assert v["gas_mixing_ratio"][2] == ["gas_species", "number_levels"]
    assert list(v["part_start_index"][0]) == list(po[:-1] + 1) and list(v["n_parts"][0]) == n_live
    assert v["aero_id"][0].dtype == np.int64 and v["aero_species"][1]["names"].split(",")[:3] == ["SO4", "NO3", "Cl"]
    clen, cstart = v["aero_component_len"][0], v["aero_component_start_ind"][0]
    assert np.array_equal(clen, state["n_orig"].sum(0)) and clen.max() > 1
    assert np.array_equal(cstart, 1 + np.concatenate([[0], np.cumsum(clen)[:-1]]))
    assert np.array_equal(v["aero_num_conc"][0], wmag[np.repeat(np.arange(4), n_live), state["wclass"]])
    back = ncio.read_column(path)
    assert np.array_equal(back["part_offset"], po) and back["next_id"] == (9 << 40) + 1
    for k in ("wclass", "ids", "least_ct", "greatest_ct", "n_orig"):
        assert np.array_equal(back[k], state[k]), k
    assert np.array_equal(back["weight_magnitude"], wmag) and back["time"] == 126.0 and back["index"] == 42
    # aero_component_* (driver:1421-1437): the lists the cells carry (two entries here), then what
    # the histogram holds beyond them; scalars have no dimension
    assert v["time"][2] == [] and v["next_id"][2] == [] and "scalar" not in dims
    tot = state["n_orig"].sum(0)
    assert (tot > 2).any() and np.array_equal(back["comp_len"], tot) and back["comp_source"].shape[1] == tot.max()
    for i in range(int(po[-1])):
        k = int(lists["comp_len"][i])
        assert np.array_equal(back["comp_source"][i, :k], lists["comp_source"][i, :k]), i
        assert np.array_equal(back["comp_time"][i, :k], lists["comp_time"][i, :k]), i
        assert (back["comp_time"][i, k:tot[i]] == state["greatest_ct"][i]).all(), i
        assert np.array_equal(np.bincount(back["comp_source"][i, :tot[i]], minlength=3), state["n_orig"][:, i]), i
    assert "create time" not in v["aero_component_create_time"][1].get("description", "")
    assert v["aero_least_create_time"][1]["description"].startswith("least (earliest) creation time")
    # mass = vol * density, vol = mass / density (:1421, init:1232): exact to one rounding each way
    np.testing.assert_allclose(back["vol"], state["vol"], rtol=3e-16, atol=0)


def test_seasalt_modes_against_the_published_formulas(oracle):
    """pmco_seasalt_modes against a numpy transcription of driver:1800-1822, 1830-1854, 1886-1914:
    single-precision literals widened to double, Re fixed at 3.1e6."""
    f32 = lambda *v: np.array(v, dtype=np.float32).astype(np.float64)
    fl, gm, gs, sc, na = oracle.seasalt_modes(1, 8.0)
    np.testing.assert_array_equal(fl, f32(73.7095947, 10.1706810))
    np.testing.assert_array_equal(gm, f32(2.03347408e-07, 1.09840811e-06) / 2)
    np.testing.assert_allclose(gs, np.log10([2.0, f32(2.2)[0]]), rtol=1e-15)
    assert abs(sc - 8.0 ** f32(3.41)[0]) <= 1e-12 * sc and na == 22.990 / (22.990 + 35.450)
    fl, gm, gs, sc, na = oracle.seasalt_modes(2, 8.0)
    coef, expon = f32(104.5, .0442, 149.6, 2.96, .51), f32(.556, 1.08, .545, .79, .87)
    off = np.array([1e5, 1e5, 1e5, 1e5, 2e5])
    np.testing.assert_allclose(fl, coef * (3.1e6 - off) ** expon, rtol=1e-14)
    np.testing.assert_array_equal(gm, .5 * f32(.018, .041, .09, .23, .83) * 1e-6)
    np.testing.assert_allclose(gs, np.log10(f32(1.37, 1.5, 1.42, 1.53, 1.85)), rtol=1e-15)
    assert sc == 1.0
    # Ovadnevaite et al. 2014 table: total flux at Re = 3.1e6 is ~2e6 m^-2 s^-1
    assert 1.5e6 < fl.sum() < 2.5e6
    assert oracle.seasalt_modes(3, 8.0)[0].size == 0


@pytest.mark.parametrize("all_columns,do_vertmix", [(True, 1), (False, 1), (True, 0)])
def test_transport_probs_against_a_numpy_transcription(oracle, all_columns, do_vertmix):
    """pmco_transport_probs (wrf_pmc_trans.F90:95-156, 205-286, 358-409, 414-562, 629-651,
    692-773) against tests/trans_fields.numpy_probs, written array-wise with numpy's matmul."""
    from tests import trans_fields as tf
    nx, nz, ny, C = 5, 12, 4, 2
    F = tf.make_fields(nx, nz, ny, C, seed=4)
    r = np.random.RandomState(1)
    n_parts = r.randint(0, 4000, (nx * nz * ny, C))
    n_parts[7] = 0                                           # an empty cell: no advective flux out
    mag = r.uniform(2e-3, 8e-3, (nx * nz * ny, C))           # some cells saturate (sum > 1)
    kw = dict(do_vertmix=do_vertmix, all_columns=all_columns, i_offset=0, j_offset=1, global_nx=nx,
              global_ny=ny + 2)
    ph, pv, ns, bad = oracle.transport_probs(nx, nz, ny, C, 60.0, 4000.0, 4000.0, F, n_parts, mag, **kw)
    rh, rv, rn = tf.numpy_probs(nx, nz, ny, C, 60.0, 4000.0, 4000.0, F, n_parts, mag, **kw)
    assert np.array_equal(ns, rn) and ns.max() > 3
    np.testing.assert_allclose(ph, rh, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(pv, rv, rtol=1e-11, atol=1e-18)
    assert not bad.any()
    tot = ph.sum(axis=2)
    k_own = (np.arange(nx * nz * ny) // nx) % nz
    off = pv.sum(axis=2) - pv[np.arange(nx * nz * ny), :, k_own]
    assert (tot + off < 1.0 + 1e-8).all() and (tot + off).max() > 0.999   # normalised cells exist
    if not all_columns:
        ncol = pv.reshape(ny, nz, nx, C, nz)
        assert not ncol[:, :, 0].any() and not ncol[:, :, nx - 1].any() and ncol[:, :, 1].any()
        assert (ns[:, 0] == 0).all() and (ns[:, 1:nx - 1] > 0).all()
    elif do_vertmix:
        rows = pv.reshape(ny, nz, nx, C, nz)[1, :, 2, 0, :]  # a column's matrix: rows sum to <= 1
        assert (rows.sum(axis=1) <= 1.0 + 1e-12).all() and (rows > 0).sum() > 3 * nz
    # a negative flux input trips the reference's assertion, reported per cell
    F2 = dict(F); F2["xkhh"] = F["xkhh"].copy(); F2["alt"] = F["alt"].copy()
    F2["alt"][2, 3, 2] = -F2["alt"][2, 3, 2]
    _, _, _, bad2 = oracle.transport_probs(nx, nz, ny, C, 60.0, 4000.0, 4000.0, F2, n_parts, mag, **kw)
    assert bad2.any()


@pytest.mark.parametrize("periodic", [True, False])
def test_transport_preweight_against_numpy(oracle, periodic):
    """pmco_transport_preweight (wrf_pmc_trans_aero.F90:1103-1423, 1428-1627) against an
    array-wise numpy form of compute_fluxes (sums in another order: equal to rounding), and the
    weight scaling where the expected count exceeds twice the ideal one."""
    nx, nz, ny, C, n_part = 4, 3, 5, 2, 300
    n = nx * nz * ny
    r = np.random.RandomState(5)
    dens = np.array([1800.0, 1200.0])
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    cells = []
    for c in range(n):
        oc = oracle.Cell(2, C, 0, 2048, dens, ve, 288.0, 1e5)
        oc.mag[:] = [3e6, 5e6]
        modes = [dict(frac=0.6, gm_radius=3e-8, gsd_log10=0.2, vol_frac=[1, 0], wclass=0, source=0),
                 dict(frac=0.4, gm_radius=1e-7, gsd_log10=0.15, vol_frac=[0, 1], wclass=1, source=0)]
        oc.generate(modes, n_part + 10 * (c % 7), 5, c, first_id=c << 32)
        cells.append(oc)
    ph = r.uniform(0.0, 0.08, (n, C, 4)); pv = r.uniform(0.0, 0.05, (n, C, nz))
    cv = r.uniform(0.7, 1.4, n)
    n_ideal = np.full((n, C), 1e9); n_ideal[3] = [40.0, 1e9]; n_ideal[17] = [1e9, 25.0]
    cnt = np.array([[int((oc.wclass[:int(oc.n_part[0])] == q).sum()) for q in range(C)] for oc in cells])
    mag0 = np.array([oc.mag.copy() for oc in cells])
    old = cnt * mag0
    ex, nr = oracle.transport_preweight(cells, nx, nz, ny, ph, pv, cv, n_ideal, 5, 0, 2, periodic, periodic)
    # numpy: losses and gains per direction
    O = old.reshape(ny, nz, nx, C); V = cv.reshape(ny, nz, nx, 1)
    PH = ph.reshape(ny, nz, nx, C, 4); PV = pv.reshape(ny, nz, nx, C, nz)
    jj, kk, ii = np.meshgrid(np.arange(ny), np.arange(nz), np.arange(nx), indexing="ij")
    if periodic:
        moves = np.ones((ny, nz, nx, 5), bool)
    else:
        inner = (ii > 0) & (ii < nx - 1) & (jj > 0) & (jj < ny - 1)
        notcorner_x = (jj > 0) & (jj < ny - 1)
        moves = np.zeros((ny, nz, nx, 5), bool)
        moves[..., :] = inner[..., None]
        moves[..., 1] |= (ii == 0) & notcorner_x
        moves[..., 0] |= (ii == nx - 1) & notcorner_x
        moves[..., 3] |= (jj == 0) & (ii > 0) & (ii < nx - 1)
        moves[..., 2] |= (jj == ny - 1) & (ii > 0) & (ii < nx - 1)
    new = O.copy()
    shifts = [(2, 1), (2, -1), (0, 1), (0, -1)]          # np.roll of the SOURCE field to its destination
    for d, (ax, sh) in enumerate([(2, -1), (2, 1), (0, -1), (0, 1)]):
        flux = PH[..., d] * O * moves[..., d:d + 1]
        new -= flux
        new += np.roll(flux * 1.0 / V, sh, axis=ax) * V   # flux * (dest_vol / source_vol)
    for k2 in range(nz):
        flux = PV[..., k2] * O * moves[..., 4:5]
        flux[:, k2] = 0.0
        new -= flux
        new[:, k2] += (flux / V).sum(axis=1) * V[:, k2]
    np.testing.assert_allclose(ex, new.reshape(n, C), rtol=1e-12)
    # scaling: only the two (cell, class) pairs with a small ideal count
    for c in range(n):
        for q in range(C):
            npn = ex[c, q] / mag0[c, q]
            if npn > 2 * n_ideal[c, q]:
                ratio = npn / n_ideal[c, q]
                assert cells[c].mag[q] == mag0[c, q] * ratio
                kept = int((cells[c].wclass[:int(cells[c].n_part[0])] == q).sum())
                assert abs(kept - cnt[c, q] / ratio) < 5 * np.sqrt(cnt[c, q] / ratio) + 3
            else:
                assert cells[c].mag[q] == mag0[c, q]
    assert nr[3] > 0 and nr[17] > 0 and nr.sum() == nr[3] + nr[17]


def test_boundary_resample_law(oracle):
    """pmco_boundary_resample (aero_state_resample, wrf_pmc_trans_aero.F90:976-1098): a class over
    its Poisson target loses exactly the excess (what stays IS the draw: mean = variance = target);
    a class short of it by less than a tenth of the target gains nothing (:1066-1079), by more it
    gains about the shortfall; a thrown-away cell is sampled anew; and the copy of a neighbour."""
    dens = np.array([1800.0])
    ve, vs = oracle.make_bin_tables(80, 1e-10, 1e-2)
    mode = dict(gm_radius=5e-8, gsd_log10=0.2, vol_frac=[1.0], wclass=0, source=0)
    def cell(n, gid):
        oc = oracle.Cell(1, 1, 1, 4096, dens, ve, 288.0, 1e5)
        oc.mag[:] = 2e6
        oc.generate([dict(mode, frac=1.0)], n, 9, gid, first_id=gid << 32)
        return oc
    rho = 1.1
    conc = lambda target: target * 2e6 / rho
    left = []
    for g in range(300):                                    # 800 particles, target 500
        oc = cell(800, g)
        ids0 = set(oc.id[:800].tolist())
        st, nid, na, nr = oc.boundary_resample(0, [dict(mode, num_conc=conc(500.0))], rho, 0.0, 9, g, 0, 1 << 40)
        n = int(oc.n_part[0])
        assert st == 0 and na == 0 and nr == 800 - n and set(oc.id[:n].tolist()) <= ids0
        left.append(n)
    left = np.array(left)
    assert abs(left.mean() - 500.0) < 4 * np.sqrt(500.0 / 300) and 0.75 < left.var() / 500.0 < 1.3
    near, far = [], []
    for g in range(200):
        oc = cell(490, g)                                   # target 500: short by ~2 % -> nothing added
        _, _, na, nr = oc.boundary_resample(0, [dict(mode, num_conc=conc(500.0))], rho, 0.0, 9, g, 1, 1 << 40)
        near.append((na, nr, int(oc.n_part[0])))
        oc = cell(300, g)                                   # short by 40 %: about 200 arrive
        _, _, na, nr = oc.boundary_resample(0, [dict(mode, num_conc=conc(500.0))], rho, 0.0, 9, g, 2, 1 << 40)
        far.append((na, nr, int(oc.n_part[0])))
    near, far = np.array(near), np.array(far)
    # (a draw above 540 is a shortfall of more than a tenth: then, and only then, particles arrive)
    assert ((near[:, 0] == 0) | (near[:, 0] > 30)).all() and (near[:, 0] == 0).mean() > 0.9
    assert (far[:, 1] == 0).all() and abs(far[:, 0].mean() - 200.0) < 6.0 and abs(far[:, 2].mean() - 500.0) < 6.0
    oc = cell(700, 7)
    st, nid, na, nr = oc.boundary_resample(1, [dict(mode, num_conc=conc(350.0))], rho, 5.0, 9, 7, 3, 1 << 40)
    n = int(oc.n_part[0])
    assert nr == 700 and na == n and abs(n - 350) < 5 * np.sqrt(350) and (oc.id[:n] >= 1 << 40).all()
    assert (oc.lct[:n] == 5.0).all() and nid == (1 << 40) + n
    src, dst = cell(333, 1), cell(50, 2)
    src.mag[:] = 7e6
    nid = dst.boundary_copy_from(src, 1 << 41)
    assert int(dst.n_part[0]) == 333 and nid == (1 << 41) + 333 and dst.mag[0] == 7e6
    assert np.array_equal(dst.vtot[:333], src.vtot[:333]) and np.array_equal(dst.id[:333], (1 << 41) + np.arange(333))


def test_independent_cdf5_walker_reads_the_assembled_column(tmp_path):
    """tests/cdf5_walker.py (a reader written from the format grammar, no code shared with ncio.py) on
    the file tests/golden/make_cdf5_column.py assembles: every dimension and variable comes back with
    the values that script states.  The -m gpu test test_column_file_against_the_oracle walks the
    DEVICE-written column with it."""
    from tests.cdf5_walker import walk
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "golden", "cdf5_column.hex")).read()
    raw = bytes.fromhex("".join(ln.split("#")[0] for ln in text.splitlines()).replace(" ", ""))
    fx = str(tmp_path / "fixture.nc")
    open(fx, "wb").write(raw)
    dims, gatts, v = walk(fx)
    assert dims == dict(aero_species=2, aero_source=2, number_levels=2, aero_particle=3, aero_components=5,
                        aero_weight_group=1, aero_weight_class=1, gas_species=1)
    assert gatts["title"].startswith("PartMC version") and len(v) == 29
    assert v["time"][2][()] == 7200.0 and v["timestep_index"][2][()] == 120
    assert v["aero_species"][1]["names"] == "SO4,BC" and v["aero_source"][1]["names"] == "traffic,sea"
    assert v["part_start_index"][2].tolist() == [1, 3] and v["n_parts"][2].tolist() == [2, 1]
    assert v["aero_particle_mass"][0] == ["aero_species", "aero_particle"]
    np.testing.assert_array_equal(v["aero_particle_mass"][2],
                                  np.array([[1e-21, 2.5e-21, 4e-22], [0.0, 5e-22, 6e-22]]) * np.array([[1800.0], [1700.0]]))
    assert v["aero_id"][2].tolist() == [7, (3 << 32) + 11, (1 << 40) + 2] and v["next_id"][2][()] == (1 << 40) + 3
    assert v["aero_component_len"][2].tolist() == [1, 3, 1] and v["aero_component_start_ind"][2].tolist() == [1, 2, 5]
    assert v["aero_component_source_num"][2].tolist() == [1, 2, 1, 1, 2]
    assert v["aero_component_create_time"][2].tolist() == [0.0, 86400.25, 12.5, 3600.0, 300.0]
    assert v["weight_magnitude"][2].reshape(-1).tolist() == [2.5e7, 1.25e7]
    assert v["gas_mixing_ratio"][2].reshape(-1).tolist() == [41.5, 38.25]
