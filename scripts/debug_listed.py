"""DEBUG: the failing many-classes case; dumps what the listed sweep of one cell took, per library build."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wrf_partmc_amd as wp
from tests import parity
from tests.test_gpu_classes import class_modes, _weights
from oracle import pmco_py as oracle
oracle.build()
n_class, n_spec = 40, 2
n_cells, n_part = 5, 3000
modes, num = class_modes(n_spec, n_class)
blk, cells, info = parity.setup_pair(wp, oracle, n_cells, n_part, n_spec, n_class, 21, 4, cap=4 * n_part,
                                     modes=modes, num_conc=num, num_conc_scale=15.0, max_appends=4096, max_events=16384)
_weights(blk, cells, n_cells, n_class, info["mag"][0, 0])
L = wp.load_library()
buf = (C.c_uint32 * (1 + 3 * 20000))()
for step in range(3):
    L.pmcx_debug_read(buf, 1, 1)
    ns, nc, st = blk.mc_coag(4, 900.0, 21, step)
    L.pmcx_debug_read(buf, 1 + 3 * 20000, 0)
    n = min(buf[0], 20000)
    a = np.frombuffer(buf, dtype=np.uint32)[1:1 + 3 * n].reshape(n, 3)
    taken = [(int(x[0]) >> 16, int(x[0]) & 0xFFFF, int(x[1]) & 1, int(x[1]) >> 16, int(x[2]) & 0xFFFF, int(x[2]) >> 16) for x in a]
    pend = [t for t in taken if t[2]]
    print(f"step {step}: gpu ns {ns.tolist()} nc {nc.tolist()} taken {n} pending {len(pend)}")
    if step == 2:
        np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"dbg_{sys.argv[1]}.npy"), a)
    for c, oc in enumerate(cells):
        r = oc.mc_coag(4, 900.0, 21, oc.cell_id, step)
        if r != (int(st[c]), int(ns[c]), int(nc[c])): print("  MISMATCH cell", c, r, (int(st[c]), int(ns[c]), int(nc[c])))
    blk.aero_state_rebalance(True, True, 21, step)
    ideal = np.full(n_class, n_part / n_class)
    for oc in cells:
        _, oc.next_id = oc.rebalance(ideal, True, True, 21, oc.cell_id, step, oc.next_id)
