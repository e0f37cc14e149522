"""ctypes host over libpmcx.so (include/pmcx.h).

Names follow the reference's call surface on the path
(interface/wrf_pmc_driver.F90:143,174,192-194,248-251): ``mc_coag``,
``aero_state_rebalance``, ``aero_state_total_particles``, ``partmc_timestep``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

COAG_KERNEL_TYPE_SEDI = 1
COAG_KERNEL_TYPE_ADDITIVE = 2
COAG_KERNEL_TYPE_CONSTANT = 3
COAG_KERNEL_TYPE_BROWN = 4
COAG_KERNEL_TYPE_ZERO = 5
COAG_KERNEL_TYPE_BROWN_SEDI = 101

# every symbol include/pmcx.h declares
EXPORTED_SYMBOLS = [
    "pmcx_config_default", "pmcx_create", "pmcx_destroy", "pmcx_last_error", "pmcx_synchronize",
    "pmcx_set_aero_data", "pmcx_set_env", "pmcx_set_weights", "pmcx_get_weights",
    "pmcx_set_n_part_ideal", "pmcx_set_weight_preparation", "pmcx_set_transport_law", "pmcx_set_next_id", "pmcx_get_next_id", "pmcx_upload_cells",
    "pmcx_get_n_part", "pmcx_download_cells", "pmcx_upload_components", "pmcx_download_components", "pmcx_update_kernel_bounds",
    "pmcx_set_kernel_bounds_margin", "pmcx_mc_coag",
    "pmcx_get_coag_counters", "pmcx_get_coag_totals", "pmcx_accumulate_totals",
    "pmcx_read_accumulated", "pmcx_rebalance", "pmcx_condense",
    "pmcx_scenario_update_aero_state", "pmcx_transport",
    "pmcx_diagnostics", "pmcx_diagnostics_ext",
    "pmcx_generate_lognormal", "pmcx_timestep", "pmcx_get_bin_tables", "pmcx_get_diag_grid",
    "pmcx_get_kmax", "pmcx_get_caches", "pmcx_sort_only", "pmcx_invalidate_sort", "pmcx_get_sort", "pmcx_test_math",
    "pmcx_test_philox", "pmcx_profile_enable", "pmcx_profile_reset", "pmcx_profile_get", "pmcx_profile_phase_cycles", "pmcx_dry_deposition",
    "pmcx_transport_begin", "pmcx_transport_export_size", "pmcx_transport_export",
    "pmcx_transport_import", "pmcx_transport_finish", "pmcx_device_alloc", "pmcx_device_free",
    "pmcx_device_copy", "pmcx_host_alloc", "pmcx_host_free", "pmcx_set_async", "pmcx_set_transport_cuts", "pmcx_transport_preweight_x_terms",
    "pmcx_memory_bytes", "pmcx_device_memory_info", "pmcx_set_cell_id_base", "pmcx_set_gas_state",
    "pmcx_get_gas_state", "pmcx_set_gas_scenario", "pmcx_scenario_update_gas_state",
    "pmcx_invalidate_lists", "pmcx_get_keys", "pmcx_seasalt_emissions",
    "pmcx_transport_probs", "pmcx_set_sampler_variant", "pmcx_transport_preweight",
    "pmcx_set_transport_lateral", "pmcx_boundary_copy", "pmcx_boundary_resample",
    "pmcx_transport_preweight_begin", "pmcx_transport_preweight_finish",
    "pmcx_get_sampler_info", "pmcx_step_flags_default", "pmcx_set_loop_cells", "pmcx_max_cap", "pmcx_set_aero_scenario_modes",
    "pmcx_grow", "pmcx_set_auto_grow", "pmcx_get_growth",
]


# Diagnostic default for new CellBlocks: (lanes, wavefronts_per_group, stamps) handed to
# pmcx_set_sampler_variant, None = the library's own choice (what production runs).
SAMPLER_VARIANT = None


class PmcxTransFields(C.Structure):
    """pmcx_trans_fields (include/pmcx.h)."""
    _fields_ = [("nx", C.c_int32), ("nz", C.c_int32), ("ny", C.c_int32), ("do_vertmix", C.c_int32),
                ("all_columns", C.c_int32), ("i_offset", C.c_int32), ("j_offset", C.c_int32),
                ("global_nx", C.c_int32), ("global_ny", C.c_int32), ("dt", C.c_double),
                ("dx", C.c_double), ("dy", C.c_double)] + \
               [(n, C.c_void_p) for n in ("alt", "xkhh", "u_flux", "v_flux", "w_flux", "z", "z_at_w",
                                          "exch_h")]


def sort_pcap(cap, n_groups):
    """Entries of a cell's slot lists (pmcx_get_sort): min(65535, cap + cap/8 + min(16 G, 8192))."""
    return min(65535, cap + cap // 8 + min(16 * n_groups, 8192))


class PmcxError(RuntimeError):
    pass


def grid_neighbours(rank, grid, periodic_x, periodic_y):
    """(below, above, west, east) ranks of `rank` in a px x py grid of blocks (rank = b * px + a), None
    where the domain ends."""
    px, py = grid
    a, b = rank % px, rank // px
    at = lambda a_, b_: (b_ % py) * px + (a_ % px)
    return (at(a, b - 1) if (periodic_y or b > 0) else None, at(a, b + 1) if (periodic_y or b < py - 1) else None,
            at(a - 1, b) if (periodic_x or a > 0) else None, at(a + 1, b) if (periodic_x or a < px - 1) else None)


def swap_sides(dist, out, nbr, extent=None):
    """The four-neighbour exchange of a domain cut in x and y.  out[side]: what goes to the neighbour
    on that side (0 below, 1 above, 2 west, 3 east; torch tensors of one dtype and device); nbr: the
    neighbours' ranks (grid_neighbours).  Returns what arrives FROM each side (None: no neighbour
    there).  Four phases: everybody sends one way and receives from the opposite side.
    extent = (real columns, real rows of this block, grid, periodic_x, periodic_y): also checks that a
    periodic direction of the whole domain is not two cells wide (check_periodic_extent)."""
    import torch
    world = dist.get_world_size()
    sizes = torch.tensor([t.numel() for t in out] + (list(extent[:2]) if extent else []), dtype=torch.int64,
                         device=out[0].device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    if extent:
        (px, py), rank = extent[2], dist.get_rank()
        a, b = rank % px, rank // px
        check_periodic_extent(sum(int(all_sizes[b * px + a_][4]) for a_ in range(px)), extent[3], "x")
        check_periodic_extent(sum(int(all_sizes[b_ * px + a][5]) for b_ in range(py)), extent[4], "y")
    got = [None] * 4
    for side in range(4):
        to, frm = nbr[side], nbr[side ^ 1]          # send towards `side`, receive from the opposite one
        ops, rbuf = [], None
        if to is not None:
            ops.append(dist.P2POp(dist.isend, out[side], to))
        if frm is not None:
            # (the sender's message is the one it addresses to ITS side `side`)
            rbuf = torch.empty(int(all_sizes[frm][side]), dtype=out[0].dtype, device=out[0].device)
            ops.append(dist.P2POp(dist.irecv, rbuf, frm))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if out[0].device.type != "cpu":
                # on RCCL wait() only orders torch's current stream behind the transfer; the library
                # unpacks on its own stream: the receive must have landed
                torch.cuda.current_stream().synchronize()
        if rbuf is not None:
            got[side ^ 1] = rbuf
    return got


def check_periodic_extent(n, periodic, name):
    """A periodic direction of a cut domain is at least three cells wide: with two, both neighbours
    of a cell are ONE cell, whose movers pmcx_transport takes as one sequence while the exchange
    delivers them as two (from below and from above) -- the arrival order, and with it the
    particles' slots, would differ from the whole domain's."""
    if periodic and n == 2:
        raise PmcxError(f"a periodic direction two cells wide cannot be cut (the domain's {name} extent is 2): "
                        "use pmcx_transport on the whole domain")


def dist_rows_before(dist, my_rows):
    """Domain rows held by the ranks below this one (ranks ordered along y)."""
    import torch
    t = torch.tensor([my_rows], dtype=torch.int64,
                     device="cpu" if dist.get_backend() != "nccl" else "cuda")
    allr = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, t)
    return int(sum(int(x.item()) for x in allr[:dist.get_rank()]))


class PmcxConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("n_cells", C.c_int32),
        ("cap", C.c_int32), ("n_spec", C.c_int32), ("n_class", C.c_int32),
        ("n_source", C.c_int32), ("n_bin", C.c_int32), ("r_min", C.c_double),
        ("r_max", C.c_double), ("max_events", C.c_int32), ("max_appends", C.c_int32),
        ("n_dbin", C.c_int32), ("i_water", C.c_int32), ("d_min", C.c_double),
        ("d_max", C.c_double), ("cell_id_base", C.c_uint32), ("compact_den", C.c_uint32),
        ("const_k", C.c_double), ("additive_beta", C.c_double),
        ("n_comp_max", C.c_int32), ("pad_", C.c_int32),
    ]


class PmcxAeroDist(C.Structure):
    _fields_ = [
        ("n_mode", C.c_int32), ("num_conc", C.c_void_p), ("gm_radius", C.c_void_p),
        ("gsd_log10", C.c_void_p), ("vol_frac", C.c_void_p), ("mode_class", C.c_void_p),
        ("mode_source", C.c_void_p), ("per_cell", C.c_int32),
    ]


class PmcxStepFlags(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("do_coagulation", C.c_int32), ("allow_doubling", C.c_int32),
        ("allow_halving", C.c_int32), ("kernel_type", C.c_int32), ("dt", C.c_double),
        ("partmc_chem_dt", C.c_double), ("elapsed_time", C.c_double),
        ("random_seed", C.c_uint64),
        ("height", C.c_void_p), ("old_height", C.c_void_p), ("aero_emission", C.c_void_p),
        ("aero_dilution_rate", C.c_void_p), ("aero_background", C.c_void_p),
        ("sea_salt_opt", C.c_int32), ("i_na", C.c_int32), ("i_cl", C.c_int32), ("pad_", C.c_int32),
        ("ocean_surface", C.c_void_p), ("wind_speed", C.c_void_p), ("ss_mode_class", C.c_void_p),
        ("ss_mode_source", C.c_void_p),
    ]


def library_path():
    return os.path.join(_HERE, "libpmcx.so")


def build_library(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    so = library_path()
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)
            if f.endswith((".hip", ".h"))] + [os.path.join(_ROOT, "include", "pmcx.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return so


_LIB = None


def load_library():
    """dlopen libpmcx.so.  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not os.path.exists(so):
        raise PmcxError(
            f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is required; there is no CPU path)")
    L = C.CDLL(so)
    L.pmcx_last_error.restype = C.c_char_p
    L.pmcx_last_error.argtypes = [C.c_void_p]
    L.pmcx_config_default.restype = None
    L.pmcx_config_default.argtypes = [C.POINTER(PmcxConfig)]
    L.pmcx_create.argtypes = [C.POINTER(PmcxConfig), C.POINTER(C.c_void_p)]
    L.pmcx_mc_coag.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_uint64, C.c_uint32]
    L.pmcx_rebalance.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32]
    L.pmcx_set_weight_preparation.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.pmcx_set_transport_law.argtypes = [C.c_void_p, C.c_int32]
    L.pmcx_set_kernel_bounds_margin.argtypes = [C.c_void_p, C.c_int32]
    L.pmcx_condense.argtypes = [C.c_void_p, C.c_double, C.c_int32]
    L.pmcx_transport.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.c_void_p, C.c_void_p]
    L.pmcx_scenario_update_aero_state.argtypes = [
        C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmcx_generate_lognormal.argtypes = [
        C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_uint64, C.c_double]
    L.pmcx_test_math.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                 C.c_void_p]
    L.pmcx_profile_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64)]
    L.pmcx_profile_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.pmcx_transport_begin.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.pmcx_transport_export_size.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    L.pmcx_transport_export.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.pmcx_transport_import.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.pmcx_transport_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmcx_device_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.pmcx_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.pmcx_host_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.pmcx_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.pmcx_set_async.argtypes = [C.c_void_p, C.c_int32]
    L.pmcx_set_transport_cuts.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.pmcx_transport_preweight_x_terms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.pmcx_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.pmcx_dry_deposition.argtypes = [C.c_void_p, C.c_double, C.c_uint64, C.c_uint32] + [C.c_void_p] * 9
    L.pmcx_timestep.argtypes = [C.c_void_p, C.POINTER(PmcxStepFlags)]
    L.pmcx_step_flags_default.restype = None
    L.pmcx_step_flags_default.argtypes = [C.POINTER(PmcxStepFlags)]
    L.pmcx_set_loop_cells.argtypes = [C.c_void_p, C.c_void_p]
    L.pmcx_set_aero_scenario_modes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.pmcx_grow.argtypes = [C.c_void_p, C.c_int32]
    L.pmcx_set_auto_grow.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double]
    L.pmcx_get_growth.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.pmcx_max_cap.restype = C.c_int32
    L.pmcx_max_cap.argtypes = [C.POINTER(PmcxConfig)]
    L.pmcx_set_sampler_variant.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.pmcx_transport_preweight_begin.argtypes = [C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p] * 4
    L.pmcx_transport_preweight_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    L.pmcx_set_transport_lateral.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.pmcx_boundary_copy.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.pmcx_boundary_resample.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_double, C.c_uint64, C.c_uint32, C.c_void_p,
                                         C.c_void_p]
    L.pmcx_transport_preweight.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_uint32, C.c_void_p, C.c_void_p]
    L.pmcx_transport_probs.argtypes = [C.c_void_p, C.POINTER(PmcxTransFields)] + [C.c_void_p] * 5
    L.pmcx_seasalt_emissions.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                                         C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    L.pmcx_get_keys.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmcx_set_cell_id_base.argtypes = [C.c_void_p, C.c_uint32]
    L.pmcx_set_gas_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.pmcx_get_gas_state.argtypes = [C.c_void_p, C.c_void_p]
    L.pmcx_set_gas_scenario.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmcx_scenario_update_gas_state.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                                 C.c_void_p]
    for name in ("pmcx_upload_cells", "pmcx_download_cells"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    for name in ("pmcx_upload_components", "pmcx_download_components"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 4
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def device_memory_info(device=0):
    """(free, total) bytes of a HIP device, through the library (no torch involved)."""
    L = load_library()
    f, t = C.c_int64(), C.c_int64()
    if L.pmcx_device_memory_info(int(device), C.byref(f), C.byref(t)) != 0:
        raise PmcxError(L.pmcx_last_error(None).decode())
    return f.value, t.value


class DeviceBuffer:
    """A plain device allocation on a block's GPU (pmcx_device_alloc): what the halo exchange of
    the transport hands around when no torch tensor is involved."""

    def __init__(self, blk, nbytes):
        self._blk, self.nbytes = blk, int(nbytes)
        self.ptr = C.c_void_p()
        blk._ck(blk._lib.pmcx_device_alloc(blk._h, self.nbytes, C.byref(self.ptr)))

    def copy_to(self, other_blk):
        """A copy owned by another block (same GPU): the in-process stand-in for a send."""
        out = DeviceBuffer(other_blk, self.nbytes)
        other_blk._ck(other_blk._lib.pmcx_device_copy(other_blk._h, out.ptr, self.ptr, self.nbytes, 0))
        return out

    def __del__(self):
        try:
            if self.ptr and self._blk._h and self._blk._h.value:
                self._blk._lib.pmcx_device_free(self._blk._h, self.ptr)
        except Exception:
            pass


class CellBlock:
    """A block of grid cells resident on one MI355X: the batched counterpart of
    the reference's ``aero_states(i,k,j)`` / ``env_states(i,k,j)`` arrays
    (WRFV3/main/module_wrf_top.F:522-525)."""

    def __init__(self, n_cells, cap, n_spec, n_class=1, n_source=0, device=0, cell_id_base=0,
                 **overrides):
        self._lib = load_library()
        cfg = PmcxConfig()
        self._lib.pmcx_config_default(C.byref(cfg))
        cfg.device, cfg.n_cells, cfg.cap, cfg.n_spec = device, n_cells, cap, n_spec
        cfg.n_class, cfg.n_source, cfg.cell_id_base = n_class, n_source, cell_id_base
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown pmcx_config field {k!r}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self._lib.pmcx_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise PmcxError(f"pmcx_create failed ({rc}): "
                            f"{self._lib.pmcx_last_error(None).decode()}")
        if SAMPLER_VARIANT is not None:     # (tests pin a variant for every block they create)
            self.set_sampler_variant(*SAMPLER_VARIANT)

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise PmcxError(f"pmcx error {rc}: {self._lib.pmcx_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for p in getattr(self, "_pinned", []):
                self._lib.pmcx_host_free(self._h, p)
            self._pinned = []
            self._lib.pmcx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        self._ck(self._lib.pmcx_synchronize(self._h))

    # -- model / environment ---------------------------------------------
    def set_aero_data(self, density, kappa=None, molec_weight=None):
        d = _f64(density)
        assert d.shape == (self.cfg.n_spec,)
        self._density = d.copy()
        self._ck(self._lib.pmcx_set_aero_data(self._h, _ptr(d), _ptr(_f64(kappa)),
                                              _ptr(_f64(molec_weight))))

    def set_env(self, temp, pressure, rel_humid=None):
        n = self.cfg.n_cells
        t = np.ascontiguousarray(np.broadcast_to(_f64(temp), (n,)))
        p = np.ascontiguousarray(np.broadcast_to(_f64(pressure), (n,)))
        r = None if rel_humid is None else np.ascontiguousarray(np.broadcast_to(_f64(rel_humid), (n,)))
        self._inflight_env = (t, p, r)
        self._ck(self._lib.pmcx_set_env(self._h, _ptr(t), _ptr(p), _ptr(r)))

    def set_weights(self, magnitude):
        m = np.broadcast_to(_f64(magnitude), (self.cfg.n_cells, self.cfg.n_class)).copy()
        self._ck(self._lib.pmcx_set_weights(self._h, _ptr(m)))

    def get_weights(self):
        m = np.zeros((self.cfg.n_cells, self.cfg.n_class))
        self._ck(self._lib.pmcx_get_weights(self._h, _ptr(m)))
        return m

    def set_n_part_ideal(self, n_ideal):
        m = np.broadcast_to(_f64(n_ideal), (self.cfg.n_cells, self.cfg.n_class)).copy()
        self._ck(self._lib.pmcx_set_n_part_ideal(self._h, _ptr(m)))

    def set_weight_preparation(self, on=True, allow_doubling=True, allow_halving=True):
        """aero_state_prepare_weight_for_add in front of every add from a distribution (emission,
        background, sea salt, the refill of an emptied boundary cell): pmcx.h.  Off by default."""
        self._ck(self._lib.pmcx_set_weight_preparation(self._h, int(on), int(allow_doubling), int(allow_halving)))

    def set_transport_law(self, law):
        """0: a mover towards a smaller weight carries 1 + geometric copies; 1: the reference's law --
        movers arrive once, particles that stay emit the other copies (pmcx.h)."""
        self._ck(self._lib.pmcx_set_transport_law(self._h, int(law)))

    def set_next_id(self, next_id):
        a = np.broadcast_to(np.asarray(next_id, dtype=np.int64), (self.cfg.n_cells,)).copy()
        self._ck(self._lib.pmcx_set_next_id(self._h, _ptr(a)))

    def get_next_id(self):
        a = np.zeros(self.cfg.n_cells, dtype=np.int64)
        self._ck(self._lib.pmcx_get_next_id(self._h, _ptr(a)))
        return a

    def set_transport_cuts(self, cut_x, nx_local=0, i_global0=0, global_nx=0):
        """The block is cut in x as well: columns 0 and nx_local - 1 are ghosts (pmcx_set_transport_cuts);
        also switches the cells' global ids (RNG keys) to rows of nx_local inside rows of global_nx."""
        self._ck(self._lib.pmcx_set_transport_cuts(self._h, int(bool(cut_x)), nx_local, i_global0, global_nx))

    def set_async(self, on=True):
        """The per-step stage calls return without waiting for the device (pmcx_set_async): results
        are valid after synchronize(); use with reuse_outputs and host_array inputs."""
        self._ck(self._lib.pmcx_set_async(self._h, int(bool(on))))

    def host_array(self, shape, dtype=np.float64):
        """A numpy array over page-locked host memory (pmcx_host_alloc) for per-step inputs and
        outputs: uploads from it are one DMA that overlaps the device's work.  Lives as long as
        the block."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        p = C.c_void_p()
        self._ck(self._lib.pmcx_host_alloc(self._h, max(n, 1) * dt.itemsize, C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_char * (max(n, 1) * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    # Set True by a caller that consumes a call's per-cell results before its next call of the same
    # kind (a time loop): the result arrays are then page-locked and reused instead of allocated.
    reuse_outputs = False

    def _out(self, key, shape, dtype=np.int32):
        if not self.reuse_outputs:
            return np.zeros(shape, dtype=dtype)
        cache = self.__dict__.setdefault("_out_cache", {})
        k = (key, tuple(np.atleast_1d(shape).tolist()), np.dtype(dtype).str)
        if k not in cache:
            cache[k] = self.host_array(shape, dtype)
        return cache[k]

    # -- particle data ----------------------------------------------------
    def upload_cells(self, cell_first, part_offset, vol, wclass=None, id=None, least_ct=None,
                     greatest_ct=None, n_orig=None):
        po = np.ascontiguousarray(part_offset, dtype=np.int64)
        n_up, nt = len(po) - 1, int(po[-1])
        v = _f64(vol)
        assert v.shape == (self.cfg.n_spec, nt), (v.shape, nt)
        wc = None if wclass is None else np.ascontiguousarray(wclass, dtype=np.int32)
        ii = None if id is None else np.ascontiguousarray(id, dtype=np.int64)
        no = None if n_orig is None else np.ascontiguousarray(n_orig, dtype=np.int32)
        self._ck(self._lib.pmcx_upload_cells(self._h, cell_first, n_up, _ptr(po), _ptr(v),
                                             _ptr(wc), _ptr(ii), _ptr(_f64(least_ct)),
                                             _ptr(_f64(greatest_ct)), _ptr(no)))

    def upload_components(self, cell_first, part_offset, comp_len, comp_source, comp_time):
        """particle%component(:) of the cells upload_cells has just laid out (driver:1421-1437):
        comp_len [n], comp_source [n][n_comp_max] (0-based), comp_time [n][n_comp_max]."""
        po = np.ascontiguousarray(part_offset, dtype=np.int64)
        K, nt = self.cfg.n_comp_max, int(po[-1])
        ln = np.ascontiguousarray(comp_len, dtype=np.int32)
        sr = np.ascontiguousarray(comp_source, dtype=np.int32)
        tm = _f64(comp_time)
        assert ln.shape == (nt,) and sr.shape == (nt, K) and tm.shape == (nt, K), (ln.shape, sr.shape, tm.shape)
        self._ck(self._lib.pmcx_upload_components(self._h, cell_first, len(po) - 1, _ptr(po), _ptr(ln),
                                                  _ptr(sr), _ptr(tm)))

    def download_components(self, cell_first=0, n_cells=None):
        """dict(part_offset, comp_len [n], comp_source [n][n_comp_max] (-1 past the end), comp_time)."""
        n_cells = self.cfg.n_cells - cell_first if n_cells is None else n_cells
        npart = self.aero_state_total_particles()[cell_first:cell_first + n_cells]
        po = np.zeros(n_cells + 1, dtype=np.int64)
        po[1:] = np.cumsum(npart)
        nt, K = int(po[-1]), self.cfg.n_comp_max
        out = dict(part_offset=po, comp_len=np.zeros(nt, dtype=np.int32),
                   comp_source=np.full((nt, K), -1, dtype=np.int32), comp_time=np.zeros((nt, K)))
        self._ck(self._lib.pmcx_download_components(self._h, cell_first, n_cells, _ptr(po), _ptr(out["comp_len"]),
                                                    _ptr(out["comp_source"]), _ptr(out["comp_time"])))
        return out

    # -- column files ------------------------------------------
    def output_column(self, prefix, ix, iy, index, cell_first, nz, species_names, time=0.0,
                      del_t=0.0, **extra):
        """Writes cells [cell_first, cell_first + nz) as one column file in the layout of
        output_column_to_file_new (wrf_pmc_driver.F90:1292-1658); returns the path."""
        from . import ncio
        dn = self.download_cells(cell_first, nz)
        if self.cfg.n_comp_max > 0:         # particle%component(:) as the device carries it (driver:1421-1437)
            cm = self.download_components(cell_first, nz)
            extra.update(comp_len=cm["comp_len"], comp_source=cm["comp_source"], comp_time=cm["comp_time"])
        path = ncio.column_filename(prefix, ix, iy, index)
        ncio.write_column(
            path, species_names=species_names, density=self._density, part_offset=dn["part_offset"],
            vol=dn["vol"], wclass=dn["wclass"], ids=dn["id"], least_ct=dn["least_ct"],
            greatest_ct=dn["greatest_ct"], n_orig=dn["n_orig"] if self.cfg.n_source > 0 else None,
            weight_magnitude=self.get_weights()[cell_first:cell_first + nz],
            next_id=int(self.get_next_id()[cell_first:cell_first + nz].max()),
            time=time, del_t=del_t, index=index, cell_ix=ix, cell_iy=iy, **extra)
        return path

    def restart_column(self, path, cell_first):
        """Loads a column file the way init_read_in_restart does (wrf_pmc_init.F90:1084-1286):
        particles, weights and next_id of cells [cell_first, cell_first + nz)."""
        from . import ncio
        col = ncio.read_column(path)
        nz = len(col["part_offset"]) - 1
        no = col["n_orig"]
        if self.cfg.n_source > 0:
            assert no.shape[0] <= self.cfg.n_source, "file has more sources than the block"
            no = np.vstack([no, np.zeros((self.cfg.n_source - no.shape[0], no.shape[1]), np.int32)])
        self.upload_cells(cell_first, col["part_offset"], col["vol"], col["wclass"], col["ids"],
                          col["least_ct"], col["greatest_ct"], no if self.cfg.n_source > 0 else None)
        if self.cfg.n_comp_max > 0:         # the lists of the file, cut at what a slot holds
            K, have = self.cfg.n_comp_max, col["comp_source"].shape[1]
            src = np.full((len(col["comp_len"]), K), -1, np.int32); tim = np.zeros((len(col["comp_len"]), K))
            src[:, :min(K, have)] = col["comp_source"][:, :K]; tim[:, :min(K, have)] = col["comp_time"][:, :K]
            self.upload_components(cell_first, col["part_offset"], np.minimum(col["comp_len"], K), src, tim)
        w = self.get_weights()
        w[cell_first:cell_first + nz] = col["weight_magnitude"]
        self.set_weights(w)
        nid = self.get_next_id()
        nid[cell_first:cell_first + nz] = col["next_id"]
        self.set_next_id(nid)
        return col

    def aero_state_total_particles(self):
        """n_part of every cell (reference: aero_state_total_particles, driver:143,174)."""
        a = np.zeros(self.cfg.n_cells, dtype=np.int32)
        self._ck(self._lib.pmcx_get_n_part(self._h, _ptr(a)))
        return a

    def download_cells(self, cell_first=0, n_cells=None):
        n_cells = self.cfg.n_cells - cell_first if n_cells is None else n_cells
        npart = self.aero_state_total_particles()[cell_first:cell_first + n_cells]
        po = np.zeros(n_cells + 1, dtype=np.int64)
        po[1:] = np.cumsum(npart)
        nt = int(po[-1])
        out = dict(
            part_offset=po,
            vol=np.zeros((self.cfg.n_spec, nt)),
            wclass=np.zeros(nt, dtype=np.int32),
            id=np.zeros(nt, dtype=np.int64),
            least_ct=np.zeros(nt), greatest_ct=np.zeros(nt),
            n_orig=np.zeros((max(self.cfg.n_source, 1), nt), dtype=np.int32),
        )
        self._ck(self._lib.pmcx_download_cells(
            self._h, cell_first, n_cells, _ptr(po), _ptr(out["vol"]), _ptr(out["wclass"]),
            _ptr(out["id"]), _ptr(out["least_ct"]), _ptr(out["greatest_ct"]),
            _ptr(out["n_orig"]) if self.cfg.n_source > 0 else None))
        return out

    def generate_lognormal(self, n_part, modes, seed, create_time=0.0):
        """modes: list of dict(frac, gm_radius, gsd_log10, vol_frac, wclass, source)."""
        fr = _f64([m["frac"] for m in modes])
        gm = _f64([m["gm_radius"] for m in modes])
        gs = _f64([m["gsd_log10"] for m in modes])
        vf = _f64([m["vol_frac"] for m in modes])
        mc = np.array([m.get("wclass", 0) for m in modes], dtype=np.int32)
        ms = np.array([m.get("source", 0) for m in modes], dtype=np.int32)
        assert vf.shape == (len(modes), self.cfg.n_spec)
        self._ck(self._lib.pmcx_generate_lognormal(self._h, n_part, len(modes), _ptr(fr), _ptr(gm),
                                                   _ptr(gs), _ptr(vf), _ptr(mc), _ptr(ms), seed,
                                                   create_time))

    # -- the path ---------------------------------------------------------
    def update_kernel_bounds(self, kernel_type):
        """Whole tables for every cell, now (est_k_minmax_binned_unweighted).  Without this call
        mc_coag keeps each cell's bounds current over its populated bins by itself."""
        self._ck(self._lib.pmcx_update_kernel_bounds(self._h, kernel_type))

    def set_kernel_bounds_margin(self, margin):
        """Bins above a cell's populated range its bounds cover (default 8); < 0: whole tables."""
        self._ck(self._lib.pmcx_set_kernel_bounds_margin(self._h, int(margin)))

    def mc_coag(self, coag_kernel_type, del_t, seed=0, step=0, counters=True):
        """mc_coag on every cell (driver:192-194).  Returns (n_samp, n_coag, status)
        arrays when counters is True (synchronises)."""
        self._ck(self._lib.pmcx_mc_coag(self._h, coag_kernel_type, del_t, seed, step))
        if counters:
            return self.coag_counters()
        return None

    def coag_counters(self):
        n = self.cfg.n_cells
        a, b, s = (self._out("coag%d" % k, n) for k in range(3))
        self._ck(self._lib.pmcx_get_coag_counters(self._h, _ptr(a), _ptr(b), _ptr(s)))
        return a, b, s

    def coag_totals(self):
        a, b, p, s = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self._lib.pmcx_get_coag_totals(self._h, C.byref(a), C.byref(b), C.byref(p),
                                                C.byref(s)))
        return a.value, b.value, p.value, s.value

    def accumulate_totals(self):
        """Add the last mc_coag's sums to the device-side running totals (asynchronous)."""
        self._ck(self._lib.pmcx_accumulate_totals(self._h))

    def read_accumulated(self, reset=True):
        a, b, p, s = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self._lib.pmcx_read_accumulated(self._h, C.byref(a), C.byref(b), C.byref(p),
                                                 C.byref(s), int(reset)))
        return a.value, b.value, p.value, s.value

    def aero_state_rebalance(self, allow_doubling=True, allow_halving=True, seed=0, step=0):
        self._ck(self._lib.pmcx_rebalance(self._h, int(allow_doubling), int(allow_halving), seed,
                                          step))

    def _aero_dist(self, modes, num_conc, n_rows=None):
        """modes: list of dict(gm_radius, gsd_log10, vol_frac, wclass, source) shared by every cell,
        or -- per cell, as scenario(i,k,j) holds them -- a list (one entry per cell) of such lists;
        num_conc: [n_cells][n_mode] (or broadcastable).  n_rows: rows of a per-cell list other than
        n_cells (pmcx_boundary_resample's list of cells)."""
        if not modes:
            return None, None
        n = self.cfg.n_cells if n_rows is None else n_rows
        if isinstance(modes, int):      # the resident modes (set_aero_scenario_modes): only num_conc travels
            keep = dict(nc=np.ascontiguousarray(np.broadcast_to(_f64(num_conc), (n, modes))))
            return PmcxAeroDist(modes, keep["nc"].ctypes.data, None, None, None, None, None, 0), keep
        per_cell = isinstance(modes[0], (list, tuple))
        if per_cell:
            assert len(modes) == n and all(len(m) == len(modes[0]) for m in modes)
            flat = [m for cell in modes for m in cell]
            M = len(modes[0])
        else:
            flat, M = modes, len(modes)
        keep = dict(
            nc=np.ascontiguousarray(np.broadcast_to(_f64(num_conc), (n, M))),
            gm=_f64([m["gm_radius"] for m in flat]), gs=_f64([m["gsd_log10"] for m in flat]),
            vf=_f64([m["vol_frac"] for m in flat]),
            mc=np.array([m.get("wclass", 0) for m in flat], dtype=np.int32),
            ms=np.array([m.get("source", 0) for m in flat], dtype=np.int32))
        d = PmcxAeroDist(M, keep["nc"].ctypes.data, keep["gm"].ctypes.data,
                         keep["gs"].ctypes.data, keep["vf"].ctypes.data, keep["mc"].ctypes.data,
                         keep["ms"].ctypes.data, int(per_cell))
        return d, keep

    def set_aero_scenario_modes(self, which, modes):
        """Keeps the mode parameters of the emission (which = 0) or background (1) distribution on
        the device (modes as for scenario_update_aero_state, shared or per cell); afterwards pass
        the number of modes (an int) instead of the modes and only num_conc travels."""
        if modes is None:
            self._ck(self._lib.pmcx_set_aero_scenario_modes(self._h, which, None))
            return
        M = len(modes[0]) if isinstance(modes[0], (list, tuple)) else len(modes)
        d, keep = self._aero_dist(modes, np.zeros((self.cfg.n_cells, M)))
        self._ck(self._lib.pmcx_set_aero_scenario_modes(self._h, which, C.addressof(d)))

    def scenario_update_aero_state(self, dt, elapsed_time, seed=0, step=0, emission=None,
                                   emission_rate=None, dilution_rate=None, background=None,
                                   background_conc=None, height=None, old_height=None):
        """scenario_update_aero_state on every cell (driver:207-210).  height, old_height:
        env_state%height now and a step ago (the dilution where the layer grew; None: unchanged).
        Returns (n_emit, n_dil_in, n_dil_out) per cell."""
        n = self.cfg.n_cells
        e, ke = self._aero_dist(emission, emission_rate)
        b, kb = self._aero_dist(background, background_conc)
        dil = None if dilution_rate is None else np.ascontiguousarray(np.broadcast_to(_f64(dilution_rate), (n,)))
        h = None if height is None or old_height is None else np.ascontiguousarray(np.broadcast_to(_f64(height), (n,)))
        ho = None if h is None else np.ascontiguousarray(np.broadcast_to(_f64(old_height), (n,)))
        self._inflight = (e, ke, b, kb, dil, h, ho)    # (an asynchronous call reads them after it has returned)
        ne, ni, no = (self._out("scen%d" % k, n) for k in range(3))
        self._ck(self._lib.pmcx_scenario_update_aero_state(
            self._h, dt, elapsed_time, seed, step, C.addressof(e) if e else None, _ptr(dil),
            C.addressof(b) if b else None, _ptr(ne), _ptr(ni), _ptr(no), _ptr(h), _ptr(ho)))
        return ne, ni, no

    def seasalt_emissions(self, dt, elapsed_time, sea_salt_opt, i_na, i_cl, ocean_surface, wind_speed,
                          height, seed=0, step=0, mode_class=None, mode_source=None):
        """seasalt_emissions on the ocean surface cells (driver:212-223, 1767-1924): option 1
        Gong (two MAM3-matched modes x wind**3.41), 2 Ovadnevaite 2014 (five modes).
        Returns (n_emit [n_cells], rate [n_cells][n_modes])."""
        n = self.cfg.n_cells
        M = 2 if sea_salt_opt == 1 else 5
        oc = np.ascontiguousarray(np.broadcast_to(ocean_surface, (n,)), dtype=np.int32)
        ws = np.ascontiguousarray(np.broadcast_to(_f64(wind_speed), (n,)))
        h = np.ascontiguousarray(np.broadcast_to(_f64(height), (n,)))
        mc = None if mode_class is None else np.ascontiguousarray(mode_class, dtype=np.int32)
        ms = None if mode_source is None else np.ascontiguousarray(mode_source, dtype=np.int32)
        ne, rate = np.zeros(n, dtype=np.int32), np.zeros((n, M))
        self._ck(self._lib.pmcx_seasalt_emissions(self._h, dt, elapsed_time, seed, step, sea_salt_opt,
                                                  i_na, i_cl, _ptr(oc), _ptr(ws), _ptr(h), _ptr(mc),
                                                  _ptr(ms), _ptr(ne), _ptr(rate)))
        return ne, rate

    # -- transport of a slab across ranks (ghost rows j = 0 and j = ny - 1) -----
    def transport_begin(self, nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=0,
                        periodic_x=True):
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        self._tr = (None if prob_h is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_h), (n, Cn, 4))),
                    None if prob_v is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_v), (n, Cn, nz))),
                    np.ascontiguousarray(np.broadcast_to(_f64(cell_volume), (n,))))
        self._ck(self._lib.pmcx_transport_begin(self._h, nx, nz, ny, int(periodic_x), _ptr(self._tr[0]),
                                                _ptr(self._tr[1]), _ptr(self._tr[2]), seed, step))

    def transport_export(self, side):
        """The movers leaving through `side` (0: towards row 0, 1: towards row ny - 1) as a
        DeviceBuffer on this block's GPU (an empty side still carries its header)."""
        nb = C.c_int64(0)
        self._ck(self._lib.pmcx_transport_export_size(self._h, side, C.byref(nb)))
        buf = DeviceBuffer(self, nb.value)
        self._ck(self._lib.pmcx_transport_export(self._h, side, buf.ptr))
        return buf

    def transport_import(self, side, buf):
        """Fills ghost row 0 (side 0) or ny - 1 (side 1) from a neighbour's export: a
        DeviceBuffer, a (device pointer, bytes) pair or a CUDA torch tensor; None: no neighbour."""
        if buf is None:
            self._ck(self._lib.pmcx_transport_import(self._h, side, None, 0))
            return
        if isinstance(buf, DeviceBuffer):
            ptr, nb = buf.ptr, buf.nbytes
        elif isinstance(buf, tuple):
            ptr, nb = C.c_void_p(buf[0]), buf[1]
        else:
            ptr, nb = C.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size()
        self._ck(self._lib.pmcx_transport_import(self._h, side, ptr, nb))

    def transport_finish(self):
        n = self.cfg.n_cells
        nm, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self._ck(self._lib.pmcx_transport_finish(self._h, _ptr(nm), _ptr(na)))
        return nm, na

    def transport_distributed(self, dist, nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=0,
                              periodic_x=True, periodic_y=True):
        """One transport step of this rank's slab inside a torch.distributed job (one process
        per GPU, ranks ordered along y): movers of the two boundary rows go to the ranks
        below and above as device buffers (RCCL point-to-point on the "nccl" backend; staged
        through the host on "gloo").  Without periodicity the end ranks have no neighbour on
        one side (the caller gives zero probabilities towards it).  Returns (n_moved,
        n_arrived) per local cell."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        self.transport_begin(nx, nz, ny, prob_h, prob_v, cell_volume, seed, step, periodic_x)
        bufs = [self.transport_export(0), self.transport_export(1)]      # to below, to above
        below = (rank - 1) % world if (periodic_y or rank > 0) else None
        above = (rank + 1) % world if (periodic_y or rank < world - 1) else None
        on_host = dist.get_backend() != "nccl"
        dev = "cpu" if on_host else f"cuda:{self.cfg.device}"
        out = []
        for bf in bufs:                                   # as tensors torch.distributed can send
            t = torch.empty(bf.nbytes, dtype=torch.uint8, device=dev)
            if bf.nbytes:
                self._ck(self._lib.pmcx_device_copy(self._h, C.c_void_p(t.data_ptr()), bf.ptr, bf.nbytes,
                                                    2 if on_host else 0))
            out.append(t)
        sizes = torch.tensor([out[0].numel(), out[1].numel(), ny - 2], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        check_periodic_extent(sum(int(x[2]) for x in all_sizes), periodic_y, "y")
        got = [None, None]                                               # from below, from above
        # phase 0: everybody sends down and receives from above; phase 1: the other way round
        for phase, (to, frm, slot) in enumerate(((below, above, 1), (above, below, 0))):
            ops, rbuf = [], None
            if to is not None:
                ops.append(dist.P2POp(dist.isend, out[phase], to))
            if frm is not None:
                rbuf = torch.empty(int(all_sizes[frm][phase]), dtype=torch.uint8, device=dev)
                ops.append(dist.P2POp(dist.irecv, rbuf, frm))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if not on_host:
                    # on RCCL wait() only orders torch's current stream behind the transfer;
                    # the library unpacks on its own stream: the receive must have landed
                    torch.cuda.current_stream().synchronize()
            if rbuf is not None:
                if on_host:
                    d = DeviceBuffer(self, rbuf.numel())
                    if rbuf.numel():
                        self._ck(self._lib.pmcx_device_copy(self._h, d.ptr, C.c_void_p(rbuf.data_ptr()),
                                                            rbuf.numel(), 1))
                    got[slot] = d
                else:
                    got[slot] = rbuf
        self.transport_import(0, got[0])
        self.transport_import(1, got[1])
        return self.transport_finish()

    # -- a domain cut in x and y over a px x py grid of ranks (rank = b * px + a): grid_neighbours,
    # -- swap_sides above
    def transport_distributed_2d(self, dist, grid, nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=0,
                                 periodic_x=True, periodic_y=True):
        """One transport step of this rank's block of a domain cut in x AND y (grid = (px, py), rank =
        b px + a; the block carries a ghost row and a ghost column on every side:
        set_transport_cuts): the movers of its four boundary strips go to the four neighbours as
        device buffers (RCCL point-to-point on "nccl", staged through the host on "gloo").  Returns
        (n_moved, n_arrived) per local cell."""
        import torch
        rank = dist.get_rank()
        nbr = grid_neighbours(rank, grid, periodic_x, periodic_y)
        self.transport_begin(nx, nz, ny, prob_h, prob_v, cell_volume, seed, step, False)
        on_host = dist.get_backend() != "nccl"
        dev = "cpu" if on_host else f"cuda:{self.cfg.device}"
        out = []
        for side in range(4):
            bf = self.transport_export(side)
            t = torch.empty(bf.nbytes, dtype=torch.uint8, device=dev)
            if bf.nbytes:
                self._ck(self._lib.pmcx_device_copy(self._h, C.c_void_p(t.data_ptr()), bf.ptr, bf.nbytes,
                                                    2 if on_host else 0))
            out.append(t)
        got = swap_sides(dist, out, nbr, extent=(nx - 2, ny - 2, grid, periodic_x, periodic_y))
        for side in range(4):
            rbuf = got[side]
            if rbuf is None:
                self.transport_import(side, None)
            elif on_host:
                d = DeviceBuffer(self, rbuf.numel())
                if rbuf.numel():
                    self._ck(self._lib.pmcx_device_copy(self._h, d.ptr, C.c_void_p(rbuf.data_ptr()), rbuf.numel(), 1))
                self.transport_import(side, d)
            else:
                self.transport_import(side, rbuf)
        return self.transport_finish()

    def transport_preweight_distributed_2d(self, dist, grid, nx, nz, ny, prob_h, prob_v, cell_volume, j_global0,
                                           global_ny, seed=0, step=0, periodic_x=True, periodic_y=True):
        """transport_preweight of this rank's block of a domain cut in x and y: the ready terms of the
        cut rows and columns travel between the four neighbours, then the boundary strips' new
        weight magnitudes for the neighbours' ghost strips.  Returns (expected_num_conc, n_removed)."""
        import torch
        rank = dist.get_rank()
        px, py = grid
        a, b = rank % px, rank // px
        nbr = grid_neighbours(rank, grid, periodic_x, periodic_y)
        dev = "cpu" if dist.get_backend() != "nccl" else f"cuda:{self.cfg.device}"
        rows, cols = self.transport_preweight_begin(nx, nz, ny, prob_h, prob_v, cell_volume, j_global0, global_ny,
                                                    periodic_x, periodic_y, cut_x=True)
        as_t = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy()).to(dev)
        got = swap_sides(dist, [as_t(rows[0]), as_t(rows[1]), as_t(cols[0]), as_t(cols[1])], nbr)
        host = lambda t: None if t is None else np.ascontiguousarray(t.cpu().numpy())
        self.transport_preweight_x_terms(host(got[2]), host(got[3]), nbr[2] is not None and a == 0,
                                         nbr[3] is not None and a == px - 1)
        ex, nr = self.transport_preweight_finish(host(got[0]), host(got[1]), nbr[0] is not None and b == 0,
                                                 nbr[1] is not None and b == py - 1, seed, step)
        # the ghost strips stand for the neighbours' boundary strips: their weights follow the scaling there
        Cn = self.cfg.n_class
        mag = self.get_weights().reshape(ny, nz, nx, Cn)
        out = [as_t(mag[1]), as_t(mag[ny - 2]), as_t(mag[:, :, 1]), as_t(mag[:, :, nx - 2])]
        got = swap_sides(dist, out, nbr)
        if got[0] is not None: mag[0] = host(got[0]).reshape(nz, nx, Cn)
        if got[1] is not None: mag[ny - 1] = host(got[1]).reshape(nz, nx, Cn)
        if got[2] is not None: mag[:, :, 0] = host(got[2]).reshape(ny, nz, Cn)
        if got[3] is not None: mag[:, :, nx - 1] = host(got[3]).reshape(ny, nz, Cn)
        self.set_weights(mag.reshape(-1, Cn))
        return ex, nr

    def dry_deposition(self, dt, aer_res_a, ustar, alpha, big_a, height, inv_density, seed=0,
                       step=0, surface=None):
        """dry_dep_aero_state on the surface cells (wrf_pmc_dep_aero.F90:154-192).
        Per-cell inputs broadcast to [n_cells]; surface: 0/1 per cell (None: every cell).
        Returns (n_removed, unrealistic) per cell."""
        n = self.cfg.n_cells
        arr = [np.ascontiguousarray(np.broadcast_to(_f64(x), (n,))) for x in
               (aer_res_a, ustar, alpha, big_a, height, inv_density)]
        sf = None if surface is None else np.ascontiguousarray(np.broadcast_to(surface, (n,)), dtype=np.int32)
        nr, bad = self._out("dep0", n), self._out("dep1", n)
        self._inflight_dep = (arr, sf)          # (an asynchronous call reads them after it has returned)
        self._ck(self._lib.pmcx_dry_deposition(self._h, dt, seed, step, _ptr(sf), *[_ptr(a) for a in arr],
                                               _ptr(nr), _ptr(bad)))
        return nr, bad

    def transport_probs(self, nx, nz, ny, dt, dx, dy, fields, do_vertmix=1, all_columns=True,
                        i_offset=0, j_offset=0, global_nx=None, global_ny=None, download=True):
        """The transport probabilities of one step from WRF's fields (float32 arrays: alt, xkhh
        [ny+2][nz][nx+2]; u_flux [n_class][ny][nz][nx+1]; v_flux [n_class][ny+1][nz][nx]; w_flux
        [n_class][ny][nz+1][nx]; z [ny][nz][nx]; z_at_w, exch_h [ny][nz+1][nx]) and the block's
        particle counts (wrf_pmc_trans.F90:95-156).  They stay on the device for the next
        transport(…, prob_h=None, prob_v=None).  download: also return (prob_h, prob_v,
        n_substeps, bad, n_class_part)."""
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        shapes = dict(alt=(ny + 2, nz, nx + 2), xkhh=(ny + 2, nz, nx + 2), u_flux=(Cn, ny, nz, nx + 1),
                      v_flux=(Cn, ny + 1, nz, nx), w_flux=(Cn, ny, nz + 1, nx), z=(ny, nz, nx),
                      z_at_w=(ny, nz + 1, nx), exch_h=(ny, nz + 1, nx))
        tf = PmcxTransFields(nx, nz, ny, int(do_vertmix), int(all_columns), i_offset, j_offset,
                             nx if global_nx is None else global_nx,
                             ny if global_ny is None else global_ny, dt, dx, dy)
        keep = []
        for name, shp in shapes.items():
            a = np.ascontiguousarray(fields[name], dtype=np.float32)
            if a.shape != shp:
                raise ValueError(f"transport_probs: {name} has shape {a.shape}, expected {shp}")
            keep.append(a)
            setattr(tf, name, a.ctypes.data)
        self._inflight_probs = keep             # (an asynchronous call reads them after it has returned)
        if not download:
            self._ck(self._lib.pmcx_transport_probs(self._h, C.byref(tf), None, None, None, None, None))
            return None
        ph, pv = np.zeros((n, Cn, 4)), np.zeros((n, Cn, nz))
        ns, bad = np.zeros((ny, nx), dtype=np.int32), np.zeros(n, dtype=np.int32)
        ncp = np.zeros((n, Cn), dtype=np.int32)
        self._ck(self._lib.pmcx_transport_probs(self._h, C.byref(tf), _ptr(ph), _ptr(pv), _ptr(ns),
                                                _ptr(bad), _ptr(ncp)))
        return ph, pv, ns, bad, ncp

    # -- open lateral boundaries (wrf_pmc_trans_aero.F90:497-534, 689-955) -----------------
    def set_transport_lateral(self, on=True, j_global0=0, global_ny=0):
        """transport() then treats the domain's lateral boundary cells as the reference's real
        case does: they only send inwards, corners nothing, what reaches them leaves the domain.
        A slab (transport_distributed) says where it lies: j_global0 = the domain row of its row 0
        (its lower ghost row), global_ny = the domain's rows."""
        self._ck(self._lib.pmcx_set_transport_lateral(self._h, int(bool(on)), j_global0, global_ny))

    def boundary_copy(self, from_cells, to_cells):
        """Outflow: every to_cell becomes a copy of its from_cell (aero_state_dup_all_particles)."""
        f = np.ascontiguousarray(from_cells, dtype=np.int32)
        t = np.ascontiguousarray(to_cells, dtype=np.int32)
        assert f.shape == t.shape
        self._ck(self._lib.pmcx_boundary_copy(self._h, len(f), _ptr(f), _ptr(t)))

    def boundary_resample(self, cells, throw_away, background, background_conc, density,
                          create_time=0.0, seed=0, step=0):
        """Inflow: aero_state_resample (wrf_pmc_trans_aero.F90:976-1098) of the listed cells
        against the background (modes: list of dicts as for scenario_update_aero_state;
        background_conc [n_list][n_mode] number mixing ratios; density [n_list]).
        Returns (n_added, n_removed) per listed cell."""
        cl = np.ascontiguousarray(cells, dtype=np.int32)
        n = len(cl)
        ta = np.ascontiguousarray(np.broadcast_to(np.asarray(throw_away, dtype=np.int32), (n,)))
        d, keep = self._aero_dist(background, background_conc, n_rows=n)
        dn = np.ascontiguousarray(np.broadcast_to(_f64(density), (n,)))
        na, nr = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self._ck(self._lib.pmcx_boundary_resample(self._h, n, _ptr(cl), _ptr(ta), C.addressof(d), _ptr(dn),
                                                  create_time, seed, step, _ptr(na), _ptr(nr)))
        return na, nr

    def lateral_boundary_conditions(self, nx, nz, ny, u_west, u_east, v_south, v_north, background,
                                    background_conc, density, seed=0, step=0, u_west_old=None,
                                    u_east_old=None, v_south_old=None, v_north_old=None,
                                    change_boundary_conditions=False, j_global0=0, global_ny=None,
                                    i_global0=0, global_nx=None):
        """The refill of the lateral boundary cells after the transport, as
        wrf_pmc_trans_aero.F90:803-955 orders it: on the west side (i = 0, rows 1 .. ny-2) a cell
        whose u (grid%u_2 at its outer face; u_west [ny][nz]) is negative copies its inner
        neighbour, otherwise it is resampled against the background -- emptied first when the
        wind has turned since the last step (u * u_old < 0) or the background has moved to its
        next time level; east (u_east: the face behind i = nx-1, outflow when positive), south
        (v_south [nx][nz], columns 1 .. nx-2) and north alike.  background_conc, density: per cell
        of the block ([n_cells][n_mode], [n_cells]); only the boundary cells' rows are used.
        A slab of a domain cut in y (rows 0 and ny - 1 ghosts) passes j_global0 (the domain row
        of its row 0) and global_ny: its west and east columns are refilled on its own real rows,
        the south / north row only by the rank that holds it (u_west, u_east indexed by local row).
        A block cut in x as well (columns 0 and nx - 1 ghosts) passes i_global0 and global_nx too: the
        west / east column is then refilled only by the rank that holds it (v_south, v_north indexed by
        local column).  Returns dict(copied=[cells], resampled=[cells], n_added, n_removed)."""
        cell = lambda i, k, j: (j * nz + k) * nx + i
        gny = ny if global_ny is None else global_ny
        slab = global_ny is not None
        rows = [j for j in (range(1, ny - 1) if slab else range(ny)) if 0 <= j + j_global0 < gny]
        x_rows = [j for j in rows if 0 < j + j_global0 < gny - 1]
        j_south = [j for j in rows if j + j_global0 == 0]
        j_north = [j for j in rows if j + j_global0 == gny - 1]
        u_w, u_e = np.asarray(u_west, float), np.asarray(u_east, float)
        v_s, v_n = np.asarray(v_south, float), np.asarray(v_north, float)
        old = [None if a is None else np.asarray(a, float)
               for a in (u_west_old, u_east_old, v_south_old, v_north_old)]
        frm, to, res, thr = [], [], [], []
        def side(cells_inner, vel, vel_old, outflow_negative):
            for (c, c_in, idx) in cells_inner:
                v = vel[idx]
                if (v < 0.0) if outflow_negative else (v > 0.0):
                    frm.append(c_in); to.append(c)
                else:
                    turned = vel_old is not None and v * vel_old[idx] < 0.0
                    res.append(c); thr.append(int(turned or change_boundary_conditions))
        gnx = nx if global_nx is None else global_nx
        cols = [i for i in (range(1, nx - 1) if global_nx is not None else range(nx)) if 0 <= i + i_global0 < gnx]
        y_cols = [i for i in cols if 0 < i + i_global0 < gnx - 1]
        for i in (i for i in cols if i + i_global0 == 0):            # the domain's west column
            side([(cell(i, k, j), cell(i + 1, k, j), (j, k)) for j in x_rows for k in range(nz)], u_w, old[0], True)
        for i in (i for i in cols if i + i_global0 == gnx - 1):      # ... and its east column
            side([(cell(i, k, j), cell(i - 1, k, j), (j, k)) for j in x_rows for k in range(nz)], u_e, old[1], False)
        for j in j_south:
            side([(cell(i, k, j), cell(i, k, j + 1), (i, k)) for i in y_cols for k in range(nz)],
                 v_s, old[2], True)
        for j in j_north:
            side([(cell(i, k, j), cell(i, k, j - 1), (i, k)) for i in y_cols for k in range(nz)],
                 v_n, old[3], False)
        if frm:
            self.boundary_copy(frm, to)
        na = nr = np.zeros(0, dtype=np.int32)
        if res:
            bc = np.broadcast_to(_f64(background_conc), (self.cfg.n_cells, len(background)))
            dn = np.broadcast_to(_f64(density), (self.cfg.n_cells,))
            na, nr = self.boundary_resample(res, thr, background, bc[res], dn[res], 0.0, seed, step)
        return dict(copied=to, copied_from=frm, resampled=res, throw_away=thr, n_added=na, n_removed=nr)

    def transport_preweight(self, nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=0,
                            periodic_x=True, periodic_y=True):
        """trans_aero_preweight (wrf_pmc_trans_aero.F90:1103-1423): expected number
        concentrations after this transport; weights scaled up where they exceed twice the ideal
        particle count.  Returns (expected_num_conc [n_cells][n_class], n_removed [n_cells])."""
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        ph = None if prob_h is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_h), (n, Cn, 4)))
        pv = None if prob_v is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_v), (n, Cn, nz)))
        cv = np.ascontiguousarray(np.broadcast_to(_f64(cell_volume), (n,)))
        ex, nr = self._out("pre0", (n, Cn), np.float64), self._out("pre1", n)
        self._inflight_pw = (ph, pv, cv)
        self._ck(self._lib.pmcx_transport_preweight(self._h, nx, nz, ny, int(periodic_x), int(periodic_y),
                                                    _ptr(ph), _ptr(pv), _ptr(cv), seed, step, _ptr(ex),
                                                    _ptr(nr)))
        return ex, nr

    def transport_preweight_begin(self, nx, nz, ny, prob_h, prob_v, cell_volume, j_global0, global_ny,
                                  periodic_x=True, periodic_y=True, cut_x=False):
        """First half of the preweight of a block with ghost rows (and, cut_x, ghost columns): returns
        the ready terms its boundary strips send out -- rows [2][nx nz][n_class] (towards row 0, towards
        row ny - 1) and, cut_x, columns [2][ny nz][n_class] (west, east), else None."""
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        ph = None if prob_h is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_h), (n, Cn, 4)))
        pv = None if prob_v is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_v), (n, Cn, nz)))
        cv = np.ascontiguousarray(np.broadcast_to(_f64(cell_volume), (n,)))
        edge = np.zeros(2 * nx * nz * Cn + (2 * ny * nz * Cn if cut_x else 0))
        self._ck(self._lib.pmcx_transport_preweight_begin(self._h, nx, nz, ny, int(periodic_x), int(periodic_y),
                                                          j_global0, global_ny, _ptr(ph), _ptr(pv), _ptr(cv),
                                                          _ptr(edge)))
        rows = edge[:2 * nx * nz * Cn].reshape(2, nx * nz, Cn)
        cols = edge[2 * nx * nz * Cn:].reshape(2, ny * nz, Cn) if cut_x else None
        return rows, cols

    def transport_preweight_x_terms(self, from_west, from_east, west_is_edge=False, east_is_edge=False):
        w = None if from_west is None else _f64(from_west)
        e = None if from_east is None else _f64(from_east)
        self._ck(self._lib.pmcx_transport_preweight_x_terms(self._h, _ptr(w), _ptr(e), int(west_is_edge),
                                                            int(east_is_edge)))

    def transport_preweight_finish(self, from_below, from_above, below_is_edge=False, above_is_edge=False,
                                   seed=0, step=0):
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        lo = None if from_below is None else _f64(from_below)
        hi = None if from_above is None else _f64(from_above)
        ex, nr = np.zeros((n, Cn)), np.zeros(n, dtype=np.int32)
        self._ck(self._lib.pmcx_transport_preweight_finish(self._h, _ptr(lo), _ptr(hi), int(below_is_edge),
                                                           int(above_is_edge), seed, step, _ptr(ex), _ptr(nr)))
        return ex, nr

    def transport_preweight_distributed(self, dist, nx, nz, ny, prob_h, prob_v, cell_volume, global_ny,
                                        seed=0, step=0, periodic_x=True, periodic_y=True):
        """transport_preweight of this rank's slab inside a torch.distributed job (ranks ordered
        along y, rows 0 and ny - 1 ghosts): the terms the cut rows send each other travel as small
        host arrays; the result on the real rows is the whole domain's, bit for bit.  The
        probabilities stay on the device: transport_distributed(…, prob_h=None, prob_v=None) may
        follow.  Returns (expected_num_conc, n_removed) per local cell."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        ph = None if prob_h is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_h), (n, Cn, 4)))
        pv = None if prob_v is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_v), (n, Cn, nz)))
        cv = np.ascontiguousarray(np.broadcast_to(_f64(cell_volume), (n,)))
        rows_before = dist_rows_before(dist, ny - 2)
        edge = np.zeros((2, nx * nz, Cn))
        self._ck(self._lib.pmcx_transport_preweight_begin(self._h, nx, nz, ny, int(periodic_x), int(periodic_y),
                                                          rows_before - 1, global_ny, _ptr(ph), _ptr(pv),
                                                          _ptr(cv), _ptr(edge)))
        below = (rank - 1) % world if (periodic_y or rank > 0) else None
        above = (rank + 1) % world if (periodic_y or rank < world - 1) else None
        dev = "cpu" if dist.get_backend() != "nccl" else f"cuda:{self.cfg.device}"
        out = [torch.from_numpy(edge[0].copy()).to(dev), torch.from_numpy(edge[1].copy()).to(dev)]
        got = [None, None]                                   # from below, from above
        for phase, (to, frm, slot) in enumerate(((below, above, 1), (above, below, 0))):
            ops, rbuf = [], None
            if to is not None:
                ops.append(dist.P2POp(dist.isend, out[phase], to))
            if frm is not None:
                rbuf = torch.empty_like(out[0])
                ops.append(dist.P2POp(dist.irecv, rbuf, frm))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if dev != "cpu":
                    torch.cuda.current_stream().synchronize()
            if rbuf is not None:
                got[slot] = np.ascontiguousarray(rbuf.cpu().numpy())
        ex, nr = np.zeros((n, Cn)), np.zeros(n, dtype=np.int32)
        self._ck(self._lib.pmcx_transport_preweight_finish(
            self._h, _ptr(got[0]), _ptr(got[1]), int(below is not None and rank == 0),
            int(above is not None and rank == world - 1), seed, step, _ptr(ex), _ptr(nr)))
        # the ghost rows stand for the neighbours' boundary rows: their weight magnitudes follow
        # the scaling done there (the reference ships the weight arrays of its ghost points,
        # wrf_pmc_trans_aero.F90:380-470)
        row = nx * nz
        mag = self.get_weights()
        out = [torch.from_numpy(mag[row:2 * row].copy()).to(dev), torch.from_numpy(mag[n - 2 * row:n - row].copy()).to(dev)]
        for phase, (to, frm, lo) in enumerate(((below, above, n - row), (above, below, 0))):
            ops, rbuf = [], None
            if to is not None:
                ops.append(dist.P2POp(dist.isend, out[phase], to))
            if frm is not None:
                rbuf = torch.empty_like(out[0])
                ops.append(dist.P2POp(dist.irecv, rbuf, frm))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if dev != "cpu":
                    torch.cuda.current_stream().synchronize()
            if rbuf is not None:
                mag[lo:lo + row] = rbuf.cpu().numpy()
        self.set_weights(mag)
        return ex, nr

    def transport(self, nx, nz, ny, prob_h, prob_v, cell_volume, seed=0, step=0, periodic_x=True,
                  periodic_y=True):
        """Stochastic transport between the block's cells (wrf_pmc_trans_aero.F90:2351-2482).
        prob_h = prob_v = None: the probabilities transport_probs left on the device.
        Returns (n_moved, n_arrived) per cell."""
        n, Cn = self.cfg.n_cells, self.cfg.n_class
        ph = None if prob_h is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_h), (n, Cn, 4)))
        pv = None if prob_v is None else np.ascontiguousarray(np.broadcast_to(_f64(prob_v), (n, Cn, nz)))
        cv = np.ascontiguousarray(np.broadcast_to(_f64(cell_volume), (n,)))
        nm, na = self._out("tr0", n), self._out("tr1", n)
        self._inflight_tr = (ph, pv, cv)
        self._ck(self._lib.pmcx_transport(self._h, nx, nz, ny, int(periodic_x), int(periodic_y),
                                          _ptr(ph), _ptr(pv), _ptr(cv), seed, step, _ptr(nm), _ptr(na)))
        return nm, na

    def condense(self, del_t, n_sub=4):
        """EXTENSION: water condensation sub-step (see include/pmcx.h)."""
        self._ck(self._lib.pmcx_condense(self._h, del_t, n_sub))

    # -- gas phase (row a9) -------------------------------------------------
    def set_cell_id_base(self, cell_id_base):
        self._ck(self._lib.pmcx_set_cell_id_base(self._h, int(cell_id_base)))
        self.cfg.cell_id_base = int(cell_id_base)

    def set_gas_state(self, mix_rat):
        """gas_state%mix_rat of every cell, [n_cells][n_gas] ppb."""
        m = np.ascontiguousarray(mix_rat, dtype=np.float64)
        assert m.ndim == 2 and m.shape[0] == self.cfg.n_cells
        self._n_gas = m.shape[1]
        self._ck(self._lib.pmcx_set_gas_state(self._h, self._n_gas, _ptr(m)))

    def get_gas_state(self):
        m = np.zeros((self.cfg.n_cells, self._n_gas))
        self._ck(self._lib.pmcx_get_gas_state(self._h, _ptr(m)))
        return m

    def set_gas_scenario(self, emission_time=None, emission_rate_scale=None, emission=None,
                         dilution_time=None, dilution_rate=None, background=None):
        """The gas half of scenario_t, kept on the device: emission [n_cells][n_time_e][n_gas]
        (mol m^-2 s^-1), dilution_rate [n_cells][n_time_d] (1/s), background
        [n_cells][n_time_d][n_gas] (ppb); either half may be left out."""
        n, g = self.cfg.n_cells, self._n_gas
        nte = 0 if emission_time is None else len(emission_time)
        ntd = 0 if dilution_time is None else len(dilution_time)
        et, es = _f64(emission_time), _f64(emission_rate_scale)
        em = None if not nte else np.ascontiguousarray(np.broadcast_to(_f64(emission), (n, nte, g)))
        dt_, dr = _f64(dilution_time), None
        bg = None
        if ntd:
            dr = np.ascontiguousarray(np.broadcast_to(_f64(dilution_rate), (n, ntd)))
            bg = np.ascontiguousarray(np.broadcast_to(_f64(background), (n, ntd, g)))
        self._ck(self._lib.pmcx_set_gas_scenario(self._h, nte, _ptr(et), _ptr(es), _ptr(em), ntd,
                                                 _ptr(dt_), _ptr(dr), _ptr(bg)))

    def scenario_update_gas_state(self, dt, elapsed_time, height, old_height=None):
        """scenario_update_gas_state on every cell (driver:202-205)."""
        n = self.cfg.n_cells
        h = np.ascontiguousarray(np.broadcast_to(_f64(height), (n,)))
        ho = None if old_height is None else np.ascontiguousarray(np.broadcast_to(_f64(old_height), (n,)))
        self._ck(self._lib.pmcx_scenario_update_gas_state(self._h, dt, elapsed_time, _ptr(h), _ptr(ho)))

    def set_loop_cells(self, in_loop):
        """The cells partmc_timestep's loop nest covers (driver:157-171): in_loop [n_cells] of 0/1,
        None: every cell.  mc_coag, the scenario updates and the rebalance leave the others alone."""
        a = None if in_loop is None else np.ascontiguousarray(in_loop, dtype=np.int32)
        assert a is None or a.shape == (self.cfg.n_cells,)
        self._ck(self._lib.pmcx_set_loop_cells(self._h, _ptr(a)))

    def max_cap(self):
        return int(self._lib.pmcx_max_cap(C.byref(self.cfg)))

    def grow(self, new_cap):
        """More slots per cell (PartMC's aero_particle_array grows; pmcx_grow): every particle, list
        and key keeps its place, no result depends on cap.  Returns the new cap."""
        self._ck(self._lib.pmcx_grow(self._h, int(new_cap)))
        return self.growth()[0]

    def set_auto_grow(self, on=True, fill=0.75, factor=1.5):
        """The library grows the block by itself (pmcx_set_auto_grow): exactly, before a transport
        gathers arrivals that would not fit; by a margin, when the fullest cell passes fill x cap at the
        start of pmcx_timestep / the boundary refill."""
        self._ck(self._lib.pmcx_set_auto_grow(self._h, int(bool(on)), float(fill), float(factor)))

    def growth(self):
        """(cap, times grown); keeps this object's cfg.cap current."""
        cap, n = C.c_int32(), C.c_int32()
        self._ck(self._lib.pmcx_get_growth(self._h, C.byref(cap), C.byref(n)))
        self.cfg.cap = cap.value
        return cap.value, n.value

    def partmc_timestep(self, elapsed_time, dt, partmc_chem_dt, do_coagulation=True,
                        allow_doubling=True, allow_halving=True,
                        kernel_type=COAG_KERNEL_TYPE_BROWN, random_seed=0, height=None,
                        old_height=None, emission=None, emission_rate=None, dilution_rate=None,
                        background=None, background_conc=None, sea_salt_opt=0, i_na=0, i_cl=0,
                        ocean_surface=None, wind_speed=None, ss_mode_class=None, ss_mode_source=None):
        """The loop body of partmc_timestep over every cell (driver:169-254): gated mc_coag,
        gas and aerosol scenario updates, sea salt, rebalance."""
        n = self.cfg.n_cells
        f = PmcxStepFlags()
        self._lib.pmcx_step_flags_default(C.byref(f))
        f.do_coagulation, f.allow_doubling, f.allow_halving = int(do_coagulation), int(allow_doubling), int(allow_halving)
        f.kernel_type, f.dt, f.partmc_chem_dt, f.elapsed_time = kernel_type, dt, partmc_chem_dt, elapsed_time
        f.random_seed = random_seed
        keep = []
        if height is not None:
            h = np.ascontiguousarray(np.broadcast_to(_f64(height), (n,))); keep.append(h)
            f.height = h.ctypes.data
            if old_height is not None:
                ho = np.ascontiguousarray(np.broadcast_to(_f64(old_height), (n,))); keep.append(ho)
                f.old_height = ho.ctypes.data
        e, ke = self._aero_dist(emission, emission_rate)
        b, kb = self._aero_dist(background, background_conc)
        if e:
            f.aero_emission = C.addressof(e)
        if b:
            f.aero_background = C.addressof(b)
        if dilution_rate is not None:
            dl = np.ascontiguousarray(np.broadcast_to(_f64(dilution_rate), (n,))); keep.append(dl)
            f.aero_dilution_rate = dl.ctypes.data
        if sea_salt_opt:
            oc = np.ascontiguousarray(np.broadcast_to(ocean_surface, (n,)), dtype=np.int32)
            ws = np.ascontiguousarray(np.broadcast_to(_f64(wind_speed), (n,)))
            keep += [oc, ws]
            f.sea_salt_opt, f.i_na, f.i_cl = int(sea_salt_opt), int(i_na), int(i_cl)
            f.ocean_surface, f.wind_speed = oc.ctypes.data, ws.ctypes.data
            if ss_mode_class is not None:
                mc = np.ascontiguousarray(ss_mode_class, dtype=np.int32); keep.append(mc)
                f.ss_mode_class = mc.ctypes.data
            if ss_mode_source is not None:
                ms = np.ascontiguousarray(ss_mode_source, dtype=np.int32); keep.append(ms)
                f.ss_mode_source = ms.ctypes.data
        self._ck(self._lib.pmcx_timestep(self._h, C.byref(f)))

    def diagnostics(self):
        """tot_num_conc, tot_mass_conc, num_dist, mass_dist (driver:838-857)."""
        n, nd = self.cfg.n_cells, self.cfg.n_dbin
        tn, tm = np.zeros(n), np.zeros(n)
        d1, d2 = np.zeros((n, nd)), np.zeros((n, nd))
        self._ck(self._lib.pmcx_diagnostics(self._h, _ptr(tn), _ptr(tm), _ptr(d1), _ptr(d2)))
        return tn, tm, d1, d2

    def diagnostics_ext(self):
        """species mass conc [n_cells][n_spec], number conc by source, PM1/2.5/10, n_parts."""
        n, S, NS = self.cfg.n_cells, self.cfg.n_spec, max(self.cfg.n_source, 1)
        sm, sn, pm = np.zeros((n, S)), np.zeros((n, NS)), np.zeros((n, 3))
        npart = np.zeros(n, dtype=np.int32)
        self._ck(self._lib.pmcx_diagnostics_ext(self._h, _ptr(sm), _ptr(sn), _ptr(pm), _ptr(npart)))
        return sm, sn[:, :self.cfg.n_source], pm, npart

    def run_part(self, kernel_type, del_t, n_steps, seed=0, rebalance=True, t_output=1):
        """Single-block time loop in the shape of PartMC's run_part (BASELINE config 1):
        per step mc_coag + rebalance; diagnostics every t_output steps."""
        hist = []
        for step in range(n_steps):
            self.mc_coag(kernel_type, del_t, seed, step, counters=False)
            if rebalance:
                self.aero_state_rebalance(True, True, seed, step)
            if (step + 1) % t_output == 0:
                hist.append(((step + 1) * del_t,) + self.diagnostics())
        return hist

    # -- introspection ----------------------------------------------------
    def bin_tables(self):
        ve, vs = np.zeros(self.cfg.n_bin + 1), np.zeros(self.cfg.n_bin * 3)
        self._ck(self._lib.pmcx_get_bin_tables(self._h, _ptr(ve), _ptr(vs)))
        return ve, vs

    def diag_grid(self):
        de, dw = np.zeros(self.cfg.n_dbin + 1), np.zeros(self.cfg.n_dbin)
        self._ck(self._lib.pmcx_get_diag_grid(self._h, _ptr(de), _ptr(dw)))
        return de, dw

    def kmax(self, cell):
        k = np.zeros((self.cfg.n_bin, self.cfg.n_bin))
        self._ck(self._lib.pmcx_get_kmax(self._h, cell, _ptr(k)))
        return k

    def caches(self, cell):
        self.growth()       # (the block may have grown: cfg.cap current)
        v, m = np.zeros(self.cfg.cap), np.zeros(self.cfg.cap)
        self._ck(self._lib.pmcx_get_caches(self._h, cell, _ptr(v), _ptr(m)))
        return v, m

    def sort_only(self):
        self._ck(self._lib.pmcx_sort_only(self._h))

    def invalidate_sort(self):
        """aero_state%valid_sort = .false. for every cell: the next mc_coag sorts them again."""
        self._ck(self._lib.pmcx_invalidate_sort(self._h))

    def set_sampler_variant(self, lanes=0, wavefronts_per_group=0, stamps=0):
        """Diagnostic: pin the pair sampler's launch shape (include/pmcx.h); 0 = automatic."""
        self._ck(self._lib.pmcx_set_sampler_variant(self._h, lanes, wavefronts_per_group, stamps))

    def sampler_info(self):
        """Cells the split-off draw of the lane-per-cell sampler left to its second launch in the
        last mc_coag (-1: that form did not run)."""
        out = (C.c_int32 * 4)()
        self._ck(self._lib.pmcx_get_sampler_info(self._h, out))
        return int(out[0])

    def sampler_form(self):
        """(form, rounds) of the last mc_coag: pmcx_set_sampler_variant's numbering (5: several weight
        classes, the screen and the sweep of its list) and, for form 5, the rounds it launched."""
        out = (C.c_int32 * 4)()
        self._ck(self._lib.pmcx_get_sampler_info(self._h, out))
        return int(out[1]), int(out[2])

    def invalidate_lists(self):
        """The lists of every cell are to be rebuilt, from sort keys that still hold."""
        self._ck(self._lib.pmcx_invalidate_lists(self._h))

    def get_keys(self, cell):
        """(gkey [cap], key_valid, evlog [n_log]) of one cell."""
        self.growth()
        k = np.zeros(self.cfg.cap, dtype=np.uint16)
        v, n = C.c_int32(), C.c_int32()
        lg = np.zeros(256, dtype=np.uint32)
        self._ck(self._lib.pmcx_get_keys(self._h, cell, _ptr(k), C.byref(v), _ptr(lg), C.byref(n)))
        return k, v.value, lg[:n.value]

    def get_sort(self, cell):
        G = self.cfg.n_class * self.cfg.n_bin
        self.growth()
        pcap = sort_pcap(self.cfg.cap, G)
        perm = np.zeros(pcap, dtype=np.uint32)
        gcnt, goff = np.zeros(G, dtype=np.int32), np.zeros(G + 1, dtype=np.int32)
        self._ck(self._lib.pmcx_get_sort(self._h, cell, _ptr(perm), _ptr(gcnt), _ptr(goff)))
        return perm, gcnt, goff

    def test_math(self, op, a, b=None, c=None, d=None, ktype=0, temp=288.15, pressure=101325.0):
        a = _f64(a)
        out = np.zeros_like(a)
        self._ck(self._lib.pmcx_test_math(self._h, op, a.size, _ptr(a), _ptr(_f64(b)),
                                          _ptr(_f64(c)), _ptr(_f64(d)), ktype, temp, pressure,
                                          _ptr(out)))
        return out

    def test_philox(self, ctr, key):
        c = np.asarray(ctr, dtype=np.uint32)
        k = np.asarray(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        self._ck(self._lib.pmcx_test_philox(self._h, _ptr(c), _ptr(k), _ptr(o)))
        return o

    def profile_enable(self, on=True):
        self._ck(self._lib.pmcx_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._ck(self._lib.pmcx_profile_reset(self._h))

    def profile_phase_cycles(self, reset=True):
        """Diagnostic (set_sampler_variant(stamps=1)): cycles per phase of the lane-per-cell sampler."""
        out = (C.c_uint64 * 12)()
        self._ck(self._lib.pmcx_profile_phase_cycles(self._h, out, 1 if reset else 0))
        return list(out)

    def profile_get(self, name):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self._lib.pmcx_profile_get(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def memory_bytes(self):
        b = C.c_int64()
        self._ck(self._lib.pmcx_memory_bytes(self._h, C.byref(b)))
        return b.value
