"""Input tables of the reference's shipped cases (data, transcribed as values).

Sources in /root/reference:
  WRFV3/test/em_scm_xy/aero_data.dat:2-21     species table
  WRFV3/test/em_scm_xy/aero_init_dist.dat:2-14  bimodal initial distribution
  WRFV3/test/em_scm_xy/aero_init_comp.dat:2-4   its composition (mass fractions)
  interface/wrf_pmc_init.F90:1764-1777          idealised log-normal IC
"""
import numpy as np

# name, density kg/m^3, ions in solution, molecular weight kg/mol, kappa
AERO_DATA_SCM = [
    ("SO4", 1770.0, 0, 96e-3, 0.65), ("NO3", 1770.0, 0, 62e-3, 0.65),
    ("Cl", 1900.0, 0, 35.5e-3, 1.28), ("NH4", 1770.0, 0, 18e-3, 0.65),
    ("MSA", 1800.0, 0, 95e-3, 0.53), ("ARO1", 1400.0, 0, 150e-3, 0.1),
    ("ARO2", 1400.0, 0, 150e-3, 0.1), ("ALK1", 1400.0, 0, 140e-3, 0.1),
    ("OLE1", 1400.0, 0, 140e-3, 0.1), ("API1", 1000.0, 0, 184e-3, 0.1),
    ("API2", 1400.0, 0, 184e-3, 0.1), ("LIM1", 1400.0, 0, 200e-3, 0.1),
    ("LIM2", 1400.0, 0, 200e-3, 0.1), ("CO3", 2600.0, 0, 60e-3, 0.53),
    ("Na", 1900.0, 0, 23e-3, 1.28), ("Ca", 2600.0, 0, 40e-3, 0.53),
    ("OIN", 2600.0, 0, 1e-3, 0.1), ("OC", 1000.0, 0, 1e-3, 0.001),
    ("BC", 1700.0, 0, 1e-3, 0.0), ("H2O", 1000.0, 0, 18e-3, 0.0),
]


def aero_data_arrays(names=None):
    rows = AERO_DATA_SCM if names is None else [r for n in names for r in AERO_DATA_SCM if r[0] == n]
    return (np.array([r[1] for r in rows]), np.array([r[4] for r in rows]),
            np.array([r[3] for r in rows]), [r[0] for r in rows])


def _vol_frac_from_mass_frac(mass_frac, names):
    """PartMC converts mass fractions to volume fractions with the densities."""
    dens, _, _, all_names = aero_data_arrays(names)
    vf = np.zeros(len(all_names))
    for sp, mf in mass_frac.items():
        i = all_names.index(sp)
        vf[i] = mf / dens[i]
    return vf / vf.sum()


def scm_init_modes(names=None):
    """em_scm_xy bimodal IC (aero_init_dist.dat) with composition aero_init_comp.dat.
    Returns (modes, total number concentration m^-3)."""
    vf = _vol_frac_from_mass_frac({"OC": 1.375, "SO4": 1.0, "NH4": 0.375}, names)
    n_small, n_large = 3.2e9, 2.9e9
    modes = [
        dict(frac=n_small, gm_radius=2e-8 / 2, gsd_log10=0.161, vol_frac=vf, wclass=0, source=0),
        dict(frac=n_large, gm_radius=1.16e-7 / 2, gsd_log10=0.217, vol_frac=vf, wclass=0, source=0),
    ]
    return modes, n_small + n_large


def idealised_modes(n_spec, n_modes=1):
    """Idealised IC of init_wrf_partmc_idealized: log-normal, char_radius 2e-8 m,
    log10_std_dev_radius 0.25, species 1 only; one weight class per mode."""
    vf = np.zeros(n_spec)
    vf[0] = 1.0
    return [dict(frac=1.0, gm_radius=2e-8, gsd_log10=0.25, vol_frac=vf, wclass=m, source=m)
            for m in range(n_modes)]


# Zhang et al. (2001) land-use parameters of the dry-deposition scheme as tabulated in
# interface/wrf_pmc_dep_aero.F90: USGS -> Zhang land-use map (:66-92), impaction parameter
# alpha (:99-102) and collector radius A [m] by (season, land use) (:104-121).  The land-use
# gamma (:94-97) is overwritten by the reference (:325) and therefore not needed.
ZHANG_FROM_USGS = (15, 7, 7, 7, 7, 10, 6, 6, 6, 6, 4, 3, 2, 1, 5, 14, 11, 10, 8, 9, 10, 9, 8, 12, 8)
ZHANG_ALPHA = (1.0, 0.6, 1.1, 0.8, 0.8, 1.2, 1.2, 50.0, 50.0, 1.3, 2.0, 50.0, 100.0, 100.0, 1.5)
ZHANG_A = (  # [land use][season]
    (0.002,) * 5, (0.005,) * 5, (0.002, 0.002, 0.005, 0.005, 0.002), (0.005, 0.005, 0.010, 0.010, 0.005),
    (0.005,) * 5, (0.002, 0.002, 0.005, 0.005, 0.002), (0.002, 0.002, 0.005, 0.005, 0.002),
    (9.999,) * 5, (9.999,) * 5, (0.010,) * 5, (0.010,) * 5, (9.999,) * 5, (9.999,) * 5, (9.999,) * 5,
    (0.010,) * 5)


def zhang_surface_parameters(ivgtyp, season=1):
    """(alpha, A) of the dry-deposition scheme for USGS vegetation types `ivgtyp`, as
    dep_aero_dry_driver selects them (wrf_pmc_dep_aero.F90:131-151: modesurftype = 2,
    iseason = 1; types outside 1..25 fall back to Zhang type 7)."""
    import numpy as np
    iv = np.atleast_1d(np.asarray(ivgtyp, dtype=np.int64))
    z = np.where((iv >= 1) & (iv <= 25), np.asarray(ZHANG_FROM_USGS)[np.clip(iv, 1, 25) - 1], 7)
    # the reference fills its real(dp) tables from default-real literals (DATA statements,
    # wrf_pmc_dep_aero.F90:94-130): the values in use are those single-precision numbers widened
    alpha = np.asarray(ZHANG_ALPHA, dtype=np.float32).astype(np.float64)[z - 1]
    big_a = np.asarray([row[season - 1] for row in ZHANG_A], dtype=np.float32).astype(np.float64)[z - 1]
    return alpha, big_a


def aerodynamic_resistance(rmol, z0, ustar, zr=2.0):
    """Aerodynamic resistance [s/m] that dry deposition receives per column: depvel_partmc
    (interface/wrf_pmc_dep.F90:188-213; called with zr = 2 m, z0 = grid%znt, rmol = 1/L the
    inverse Monin-Obukhov length, :85-91).  Host-side, one value per column."""
    import numpy as np
    rmol = np.where(np.abs(np.asarray(rmol, dtype=np.float64)) < 1.0e-6, 0.0, rmol)
    z0 = np.asarray(z0, dtype=np.float64)
    ustar = np.asarray(ustar, dtype=np.float64)
    neutral = 0.74 * np.log(zr / z0)
    with np.errstate(invalid="ignore", divide="ignore"):
        ar = ((1.0 - 9.0 * zr * rmol) ** 0.25 + 0.001) ** 2
        ao = ((1.0 - 9.0 * z0 * rmol) ** 0.25 + 0.001) ** 2
        unstable = 0.74 * (np.log((ar - 1.0) / (ar + 1.0)) - np.log((ao - 1.0) / (ao + 1.0)))
    stable = neutral + 4.7 * rmol * (zr - z0)
    polint = np.where(rmol < 0.0, unstable, np.where(rmol == 0.0, neutral, stable))
    return polint / (0.40 * np.maximum(ustar, 1.0e-4))
