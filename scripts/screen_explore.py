import numpy as np, sys
sys.path.insert(0, '.')
import wrf_partmc_amd as wp
r = np.logspace(-10, -4, 61)
with wp.CellBlock(1, 64, 1) as blk:
    for T, p in ((288.15, 101325.0), (220.0, 25000.0)):
        for dens in (1000.0, 2600.0):
            print("T", T, "p", p, "dens", dens)
            # pairs (r, r) and (r, 10 r) and (r, 1e-7)
            for lab, r2 in (("same", r), ("x10", np.minimum(r * 10, 1e-4)), ("fix3e-8", np.full_like(r, 3e-8))):
                v1, v2 = 4 / 3 * np.pi * r ** 3, 4 / 3 * np.pi * r2 ** 3
                k64 = blk.test_math(7, v1, v1 * dens, v2, v2 * dens, ktype=4, temp=T, pressure=p)
                k32g = blk.test_math(9, v1, v1 * dens, v2, v2 * dens, ktype=4, temp=T, pressure=p)
                k32 = blk.test_math(9, v1, v1 * dens, v2, v2 * dens, ktype=1000, temp=T, pressure=p)
                err = np.abs(k32 / k64 - 1)
                print(lab, " ".join(f"{ri:.0e}:{'Y' if g > 0 else 'n'}{e:.0e}" for ri, g, e in zip(r[::3], k32g[::3], err[::3])))
