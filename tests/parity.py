"""Shared harness: run the HIP path and the CPU oracle on identical inputs and
compare every particle field bit for bit.  Used by tests/, smoke() and bench."""
import numpy as np

DIAM_TOL = 1e-12  # relative tolerance on fp64 tree-summed mass diagnostics


def default_species(n_spec):
    from wrf_partmc_amd.data import AERO_DATA_SCM
    order = ["SO4", "BC", "OC", "NH4", "NO3", "Na", "Cl", "OIN", "H2O"]
    names = [n for n in order][:n_spec]
    if n_spec > len(order):
        names = [r[0] for r in AERO_DATA_SCM][:n_spec]
    dens = np.array([next(r[1] for r in AERO_DATA_SCM if r[0] == n) for n in names])
    return names, dens


def make_modes(n_spec, n_class, coarse=False):
    """Bimodal em_scm_xy-like population (aero_init_dist.dat) spread over classes."""
    rng = np.random.RandomState(1234)
    modes = []
    base = [(3.2e9, 1e-8, 0.161), (2.9e9, 5.8e-8, 0.217)]
    if coarse:
        base.append((2.0e7, 1.5e-6, 0.25))
    for k, (num, gm, gsd) in enumerate(base):
        vf = rng.uniform(0.1, 1.0, n_spec)
        vf /= vf.sum()
        modes.append(dict(frac=num, gm_radius=gm, gsd_log10=gsd, vol_frac=vf,
                          wclass=k % n_class, source=k))
    return modes, sum(b[0] for b in base)


# particle%component(:) entries carried by the blocks setup_pair builds (tests/test_gpu_components.py
# runs the parity tests again with it set): every comparison then covers the lists as well
N_COMP_MAX = 0


def upload_components(blk, cells, po):
    """The oracle cells' component lists to the block (after upload_cells with the same offsets)."""
    K, nt = blk.cfg.n_comp_max, int(po[-1])
    ln = np.zeros(nt, np.int32); sr = np.full((nt, K), -1, np.int32); tm = np.zeros((nt, K))
    for c, oc in enumerate(cells):
        s = slice(po[c], po[c + 1])
        n = po[c + 1] - po[c]
        ln[s] = oc.comp_n[:n]; sr[s] = oc.comp_src[:n]; tm[s] = oc.comp_time[:n]
    blk.upload_components(0, po, ln, sr, tm)


def setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, seed, kernel, cap=None,
               n_source=3, coarse=False, cell_id_base=0, max_events=None, max_appends=None,
               num_conc_scale=1.0, device=0, explicit_bounds=True, modes=None, num_conc=None, **cfg_over):
    """Build a CellBlock and matching oracle cells with identical contents."""
    cap = cap or int(np.ceil(n_part * 1.0 / 64) * 64)
    over = dict(cfg_over)
    if N_COMP_MAX and n_source > 0:
        over.setdefault("n_comp_max", N_COMP_MAX)
    kc = over.get("n_comp_max", 0)
    if max_events is not None:
        over["max_events"] = max_events
    if max_appends is not None:
        over["max_appends"] = max_appends
    blk = wp.CellBlock(n_cells, cap, n_spec, n_class=n_class, n_source=n_source, device=device,
                       cell_id_base=cell_id_base, **over)
    names, dens = default_species(n_spec)
    blk.set_aero_data(dens)
    temp = 288.15 - 6.5 * np.arange(n_cells) % 60
    pres = 101325.0 * (temp / 288.15) ** 5.256
    blk.set_env(temp, pres, np.full(n_cells, 0.5))
    v_edge, v_samp = blk.bin_tables()
    if modes is None:
        modes, num_conc = make_modes(n_spec, n_class, coarse)
    num_conc *= num_conc_scale
    # flat weights: every computational particle of class c stands for mag[c] m^-3
    frac_cls = np.zeros(n_class)
    for m in modes:
        frac_cls[m["wclass"]] += m["frac"]
    mag = np.zeros((n_cells, n_class))
    for c in range(n_cells):
        # classes get different magnitudes on purpose (weighted coagulation)
        mag[c] = num_conc / n_part * (1.0 + 0.5 * np.arange(n_class))
    blk.set_weights(mag)
    blk.set_n_part_ideal(np.full((n_cells, n_class), n_part / n_class))
    blk.set_next_id((np.arange(n_cells, dtype=np.int64) + cell_id_base << 32) + (1 << 24))

    cells = []
    po = np.zeros(n_cells + 1, dtype=np.int64)
    for c in range(n_cells):
        oc = oracle.Cell(n_spec, n_class, n_source, cap, dens, v_edge, temp[c], pres[c],
                         max_events=blk.cfg.max_events, max_appends=blk.cfg.max_appends, n_comp_max=kc)
        oc.mag[:] = mag[c]
        gid = cell_id_base + c
        oc.generate(modes, n_part, seed, gid, first_id=gid << 32, create_time=float(c))
        if kc:      # every particle its own create time, so that a list shows whose entries it holds
            oc.lct[:n_part] = oc.gct[:n_part] = oc.comp_time[:n_part, 0] = c + 1e-4 * np.arange(n_part)
        oc.cell_id = gid
        oc.next_id = int((gid << 32) + (1 << 24))
        cells.append(oc)
        po[c + 1] = po[c] + n_part
    nt = int(po[-1])
    vol = np.zeros((n_spec, nt)); wc = np.zeros(nt, np.int32); ids = np.zeros(nt, np.int64)
    lct = np.zeros(nt); gct = np.zeros(nt); norig = np.zeros((max(n_source, 1), nt), np.int32)
    for c, oc in enumerate(cells):
        s = slice(po[c], po[c + 1])
        vol[:, s] = oc.vol[:, :n_part]; wc[s] = oc.wclass[:n_part]; ids[s] = oc.id[:n_part]
        lct[s] = oc.lct[:n_part]; gct[s] = oc.gct[:n_part]; norig[:, s] = oc.n_orig[:, :n_part]
    blk.upload_cells(0, po, vol, wc, ids, lct, gct, norig if n_source > 0 else None)
    if kc:
        upload_components(blk, cells, po)
    if explicit_bounds:      # whole tables now; otherwise mc_coag computes what each cell needs
        blk.update_kernel_bounds(kernel)
    for c, oc in enumerate(cells):
        oc.kmax[:] = oracle.kmax_table(kernel, v_samp, dens.min(), dens.max(), temp[c], pres[c],
                                       blk.cfg.const_k, blk.cfg.additive_beta)
    return blk, cells, dict(modes=modes, mag=mag, temp=temp, pres=pres, dens=dens, v_edge=v_edge,
                            v_samp=v_samp)


def compare_state(blk, cells, check_caches=True):
    """Returns (equal, message).  Bitwise on every particle field."""
    dn = blk.download_cells()
    po = dn["part_offset"]
    comps = None
    for c, oc in enumerate(cells):
        n = int(oc.n_part[0])
        if po[c + 1] - po[c] != n:
            return False, f"cell {c}: n_part gpu {po[c+1]-po[c]} oracle {n}"
        s = slice(po[c], po[c + 1])
        for name, a, b in [
            ("vol", dn["vol"][:, s], oc.vol[:, :n]), ("wclass", dn["wclass"][s], oc.wclass[:n]),
            ("id", dn["id"][s], oc.id[:n]), ("least_ct", dn["least_ct"][s], oc.lct[:n]),
            ("greatest_ct", dn["greatest_ct"][s], oc.gct[:n]),
        ] + ([("n_orig", dn["n_orig"][:, s], oc.n_orig[:, :n])] if blk.cfg.n_source > 0 else []):
            if not np.array_equal(np.ascontiguousarray(a).view(np.uint8),
                                  np.ascontiguousarray(b).view(np.uint8)):
                bad = np.argwhere(np.asarray(a) != np.asarray(b))
                return False, f"cell {c}: field {name} differs at {bad[:4].tolist()} ({len(bad)} entries)"
        if blk.cfg.n_comp_max > 0:
            if oc.n_comp_max != blk.cfg.n_comp_max:
                return False, f"cell {c}: the oracle cell was built without component lists"
            if comps is None:
                comps = blk.download_components()
            for name, a, b in [("comp_len", comps["comp_len"][s], oc.comp_n[:n]),
                               ("comp_source", comps["comp_source"][s], oc.comp_src[:n]),
                               ("comp_time", comps["comp_time"][s], oc.comp_time[:n])]:
                if not np.array_equal(np.ascontiguousarray(a).view(np.uint8),
                                      np.ascontiguousarray(b).view(np.uint8)):
                    bad = np.argwhere(np.asarray(a) != np.asarray(b))
                    return False, f"cell {c}: field {name} differs at {bad[:4].tolist()} ({len(bad)} entries)"
        if check_caches:
            vt, ms = blk.caches(c)
            if not (np.array_equal(vt[:n], oc.vtot[:n]) and np.array_equal(ms[:n], oc.mass[:n])):
                return False, f"cell {c}: vtot/mass caches differ"
    return True, ""


def run_pair(n_cells=4, n_part=1000, n_spec=3, n_class=1, steps=2, seed=0, kernel=4, del_t=300.0,
             rebalance=False, **kw):
    import wrf_partmc_amd as wp
    from oracle import pmco_py as oracle
    blk, cells, info = setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, seed, kernel, **kw)
    res = dict(n_cells=n_cells, bitwise_equal=True, first_mismatch="", n_coag_total=0,
               n_samp_total=0, status_or=0)
    kmax0 = blk.kmax(0)
    if not np.array_equal(kmax0, cells[0].kmax):
        res.update(bitwise_equal=False, first_mismatch="kmax table of cell 0 differs")
        return res
    for step in range(steps):
        ns, nc, st = blk.mc_coag(kernel, del_t, seed, step)
        for c, oc in enumerate(cells):
            ost, ons, onc = oc.mc_coag(kernel, del_t, seed, oc.cell_id, step, blk.cfg.const_k,
                                       blk.cfg.additive_beta)
            if (ons, onc, ost) != (int(ns[c]), int(nc[c]), int(st[c])):
                res.update(bitwise_equal=False, first_mismatch=(
                    f"step {step} cell {c}: (n_samp,n_coag,status) gpu "
                    f"{(int(ns[c]), int(nc[c]), int(st[c]))} oracle {(ons, onc, ost)}"))
                return res
        res["n_coag_total"] += int(nc.sum()); res["n_samp_total"] += int(ns.sum())
        res["status_or"] |= int(np.bitwise_or.reduce(st))
        ok, msg = compare_state(blk, cells)
        if not ok:
            res.update(bitwise_equal=False, first_mismatch=f"step {step} after mc_coag: {msg}")
            return res
        if rebalance:
            blk.aero_state_rebalance(True, True, seed, step)
            for oc in cells:
                _, oc.next_id = oc.rebalance(np.full(n_class, n_part / n_class), True, True, seed,
                                             oc.cell_id, step, oc.next_id)
            ok, msg = compare_state(blk, cells)
            if ok and not np.array_equal(blk.get_weights(), np.array([oc.mag for oc in cells])):
                ok, msg = False, "weight magnitudes differ"
            if not ok:
                res.update(bitwise_equal=False, first_mismatch=f"step {step} after rebalance: {msg}")
                return res
    res["blk"], res["cells"], res["info"] = blk, cells, info
    return res
