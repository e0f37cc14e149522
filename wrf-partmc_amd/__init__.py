"""wrf-partmc_amd: MI355X-native PartMC per-cell microphysics (coagulation path).

Python host side of the C-ABI in include/pmcx.h.  It mirrors the names the
WRF-PartMC interface uses on this path (interface/wrf_pmc_driver.F90):
``mc_coag``, ``aero_state_rebalance``, ``aero_state_total_particles``,
``partmc_timestep`` -- batched over all cells a context owns.

There is no CPU fallback: importing works anywhere, but creating a
:class:`CellBlock` needs libpmcx.so (built by ``__graft_entry__.build()``)
and a gfx950 device, and fails loudly otherwise.
"""
from .host import (  # noqa: F401
    COAG_KERNEL_TYPE_ADDITIVE,
    COAG_KERNEL_TYPE_BROWN,
    COAG_KERNEL_TYPE_BROWN_SEDI,
    COAG_KERNEL_TYPE_CONSTANT,
    COAG_KERNEL_TYPE_SEDI,
    COAG_KERNEL_TYPE_ZERO,
    CellBlock,
    PmcxConfig,
    PmcxError,
    PmcxAeroDist,
    PmcxStepFlags,
    build_library,
    device_memory_info,
    library_path,
    load_library,
    EXPORTED_SYMBOLS,
)
from .data import AERO_DATA_SCM, idealised_modes, scm_init_modes  # noqa: F401
