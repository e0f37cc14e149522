"""Synthetic WRF fields for the transport-probability build, and an independent numpy
transcription of the reference's routines (interface/wrf_pmc_trans.F90:205-286, 358-409,
414-562, 629-651, 692-773) written array-wise -- not the oracle's loops -- to pin the oracle."""
import numpy as np


def make_fields(nx, nz, ny, n_class, seed=0, k_scale=30.0, flux_scale=1.0):
    """A plausible atmosphere: stretched levels, decaying density, a convective K profile,
    mass fluxes of either sign.  float32, shaped as oracle.TRANS_FIELD_SHAPES."""
    r = np.random.RandomState(seed)
    f32 = np.float32
    zw1 = np.concatenate([[0.0], np.cumsum(40.0 * 1.12 ** np.arange(nz))])           # [nz+1]
    ter = r.uniform(0.0, 30.0, (ny, 1, nx))
    z_at_w = (zw1[None, :, None] + ter).astype(f32)
    z = (0.5 * (z_at_w[:, 1:, :].astype(np.float64) + z_at_w[:, :-1, :])).astype(f32)
    zc = 0.5 * (zw1[1:] + zw1[:-1])
    alt = (1.0 / (1.2 * np.exp(-zc / 8000.0)))[None, :, None] * r.uniform(0.97, 1.03, (ny + 2, nz, nx + 2))
    xkhh = r.uniform(0.0, 400.0, (ny + 2, nz, nx + 2))
    exch = k_scale * (zw1 / 800.0) * np.exp(1.0 - zw1 / 800.0)
    exch_h = exch[None, :, None] * r.uniform(0.2, 1.5, (ny, nz + 1, nx))
    exch_h[:, :, 0] *= 1e-9                                                           # a calm column
    u = flux_scale * r.normal(0.0, 1.0, (n_class, ny, nz, nx + 1))
    v = flux_scale * r.normal(0.0, 1.0, (n_class, ny + 1, nz, nx))
    w = flux_scale * r.normal(0.0, 0.3, (n_class, ny, nz + 1, nx))
    w[:, :, 0, :] = 0.0
    w[:, :, nz, :] = 0.0
    return dict(alt=alt.astype(f32), xkhh=xkhh.astype(f32), u_flux=u.astype(f32), v_flux=v.astype(f32),
                w_flux=w.astype(f32), z=z, z_at_w=z_at_w, exch_h=exch_h.astype(f32))


def numpy_probs(nx, nz, ny, C, dt, dx, dy, F, n_parts, mag, do_vertmix=1, all_columns=True,
                i_offset=0, j_offset=0, global_nx=None, global_ny=None):
    """Array-wise transcription; matrix powers with numpy's matmul (so: equal to rounding)."""
    gx = nx if global_nx is None else global_nx
    gy = ny if global_ny is None else global_ny
    d = np.float64
    alt = F["alt"].astype(d)
    rho = 1.0 / alt                                               # [ny+2][nz][nx+2]
    rho_c = rho[1:-1, :, 1:-1]
    npart = np.asarray(n_parts, d).reshape(ny, nz, nx, C)
    mg = np.asarray(mag, d).reshape(ny, nz, nx, C)
    mix = np.moveaxis(npart * mg, 3, 0) / rho_c[None]            # [C][ny][nz][nx]
    u, v, w = F["u_flux"].astype(d), F["v_flux"].astype(d), F["w_flux"].astype(d)
    with np.errstate(divide="ignore", invalid="ignore"):
        pos = lambda fl: np.where((fl > 0) & (mix > 0), fl / mix, 0.0)
        neg = lambda fl: np.where((fl < 0) & (mix > 0), -fl / mix, 0.0)
        a_xp, a_xm = pos(u[:, :, :, 1:]), neg(u[:, :, :, :-1])
        a_yp, a_ym = pos(v[:, 1:, :, :]), neg(v[:, :-1, :, :])
        a_zp, a_zm = np.minimum(pos(w[:, :, 1:, :]), .95), np.minimum(neg(w[:, :, :-1, :]), .95)
    adv = np.stack([a_xp, a_xm, a_yp, a_ym, a_zp, a_zm])         # [6][C][ny][nz][nx]
    ps = adv.sum(axis=0)
    adv = np.where(ps > 1.0, adv / (ps + 1e-8), adv)
    xk = F["xkhh"]                                                # float32 sums, as the reference
    kc = xk[1:-1, :, 1:-1]
    def dif(sl_j, sl_i, h):
        dc = np.maximum(.5 * (kc + xk[sl_j, :, sl_i]).astype(d), 0.0)
        re = .5 * (rho_c + rho[sl_j, :, sl_i])
        return dt * dc * re / (h * h * rho_c)
    c_ = slice(1, -1)
    dif4 = np.stack([dif(c_, slice(2, None), dx), dif(c_, slice(0, -2), dx),
                     dif(slice(2, None), c_, dy), dif(slice(0, -2), c_, dy)])   # x+, x-, y+, y-
    pv = np.zeros((C, ny, nz, nx, nz))
    nsub = np.zeros((ny, nx), np.int32)
    z, zw, ex = F["z"], F["z_at_w"].astype(d), F["exch_h"]
    for j in range(ny):
        for i in range(nx):
            gi, gj = i + i_offset + 1, j + j_offset + 1
            if not all_columns and (gi < 2 or gi > gx - 1 or gj < 2 or gj > gy - 1):
                continue
            min_dz = d(z[j, 1, i] - z[j, 0, i])
            max_k = max(d(ex[j, :, i].max()), 1e-6)
            n = int(np.ceil(dt / (.1 * min_dz ** 2 / max_k)))
            sdt = dt / n
            nsub[j, i] = n
            zm, rm, zwc = z[j, :, i].astype(d), rho_c[j, :, i], zw[j, :, i]
            kw = np.maximum(ex[j, 1:nz, i].astype(d), 1e-6)                       # faces 1..nz-1
            dzm = zm[1:] - zm[:-1]
            rw = (zm[1:] - zwc[1:nz]) / dzm * rm[:-1] + (zwc[1:nz] - zm[:-1]) / dzm * rm[1:]
            dzw = zwc[1:] - zwc[:-1]
            up = np.zeros(nz); dn = np.zeros(nz)
            up[:-1] = sdt / (dzw[:-1] * dzm) * (rw / rm[:-1]) * kw
            dn[1:] = sdt / (dzw[1:] * dzm) * (rw / rm[1:]) * kw
            B = np.diag(1.0 - up - dn) + np.diag(up[:-1], 1) + np.diag(dn[1:], -1)
            Bt = B.copy()
            for _ in range(2, n + 1):
                Bt = Bt @ B
                Bt /= Bt.sum(axis=1, keepdims=True)
            for cl in range(C):
                zp, zmn = adv[4, cl, j, :, i].copy(), adv[5, cl, j, :, i].copy()
                zp[-1] = 0.0; zmn[0] = 0.0
                A = np.diag(1.0 - (zp + zmn)) + np.diag(zp[:-1], 1) + np.diag(zmn[1:], -1)
                pv[cl, j, :, i, :] = Bt @ A if do_vertmix > 0 else A
    own = np.eye(nz, dtype=bool)[None, None, :, None, :]
    off = np.where(own, 0.0, pv).sum(axis=-1)                     # [C][ny][nz][nx]
    tot = adv[:4].sum(axis=0) + dif4.sum(axis=0)[None] + off
    s = np.where(tot >= 1.0, tot, 1.0)
    ph = np.stack([adv[1] + dif4[1][None], adv[0] + dif4[0][None],
                   adv[3] + dif4[3][None], adv[2] + dif4[2][None]], axis=-1) / s[..., None]
    pv = pv / s[..., None]
    n_cells = nx * nz * ny
    return (np.moveaxis(ph, 0, 3).reshape(n_cells, C, 4), np.moveaxis(pv, 0, 3).reshape(n_cells, C, nz), nsub)
