"""Column NetCDF particle files of WRF-PartMC: writer and restart reader.

The reference writes one file per (i, j) column and output time,
`<prefix>_<ix:03d>_<iy:03d>_<index:08d>.nc`, from `output_column_to_file_new`
(interface/wrf_pmc_driver.F90:1292-1658; variable names, dimensions, units at :1508-1653)
and reads it back in `init_read_in_restart` (interface/wrf_pmc_init.F90:1084-1286).
`aero_id` and `next_id` are 64-bit integers, which the classic NetCDF formats (CDF-1/2)
cannot hold; the files are therefore written in the CDF-5 ("64-bit data") variant of the
classic format, which any netCDF >= 4.4 opens.  No netCDF library exists in this image, so
the encoder/decoder below is self-contained (the file format is short: header of dimension,
attribute and variable lists, then the big-endian arrays, each padded to 4 bytes); it also
speaks CDF-2, which is how it is cross-checked against `scipy.io.netcdf_file` in the tests.

The `aero_component_*` records are the particles' own lists (`particle%component(:)`, driver:1421-1437)
when the block carries them (`pmcx_config.n_comp_max`, CellBlock.download_components); constituents a
particle has beyond the entries carried -- or all of them without lists -- are expanded from the
per-source histogram `n_orig` in source order, the first created at the particle's least create
time and the others at its greatest.  Gas mixing ratios are passed through as given (MOSAIC is out
of scope), optical properties are not written (`record_optical = .false.`).

Container [UPSTREAM-UNCITED]: which format `pmc_nc_open_write` creates is decided in PartMC's netcdf
module (absent submodule).  `aero_id` and `next_id` are written with `pmc_nc_write_integer64*`
(driver:1553,1652), i.e. NF90_INT64, which the classic CDF-1/2 formats cannot hold: the reference's
files are therefore either CDF-5 (NF90_64BIT_DATA) or netCDF-4/HDF5 (NF90_NETCDF4).  This module
writes and reads CDF-5; a netCDF-4 file is recognised by its HDF5 signature and refused with the
conversion that makes it readable (`nccopy -k cdf5 in.nc out.nc`); any netCDF library >= 4.4 opens
what is written here.  Scalars (`time`, `timestep`, `timestep_index`, `next_id`) are variables
without dimensions, as `pmc_nc_write_real` / `_integer` / `_integer64` define them upstream; the
reader also accepts them with a dimension of length one.
"""
import struct

import numpy as np

NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE = 1, 2, 3, 4, 5, 6
NC_INT64 = 10
_TAG_DIM, _TAG_VAR, _TAG_ATT = 0x0A, 0x0B, 0x0C
_DTYPES = {NC_BYTE: ">i1", NC_CHAR: "S1", NC_SHORT: ">i2", NC_INT: ">i4", NC_FLOAT: ">f4",
           NC_DOUBLE: ">f8", NC_INT64: ">i8"}


def _nc_type(arr):
    k = arr.dtype.kind
    if k in "SU":
        return NC_CHAR
    if k == "f":
        return NC_DOUBLE if arr.dtype.itemsize == 8 else NC_FLOAT
    if k in "iub":
        return NC_INT64 if arr.dtype.itemsize == 8 else NC_INT
    raise TypeError(f"unsupported dtype {arr.dtype}")


def _pad4(b):
    return b + b"\x00" * (-len(b) % 4)


class CdfWriter:
    """Minimal writer of classic-model NetCDF files (version 2 = 64-bit offset, 5 = 64-bit data),
    fixed-size dimensions only."""

    def __init__(self, version=5):
        assert version in (2, 5)
        self.version = version
        self.dims, self.gatts, self.vars = [], [], []

    def dim(self, name, size):
        self.dims.append((name, int(size)))
        return len(self.dims) - 1

    def att(self, name, value):
        self.gatts.append((name, value))

    def var(self, name, dims, data, nc_type=None, **atts):
        """dims: names in file (C) order, slowest first."""
        data = np.asarray(data)
        ids = [[d[0] for d in self.dims].index(n) for n in dims]
        shape = tuple(self.dims[i][1] for i in ids)
        assert data.size == int(np.prod(shape)), (name, data.shape, shape)
        t = nc_type or _nc_type(data)
        self.vars.append((name, ids, np.ascontiguousarray(data.reshape(shape)), t, atts))

    # -- encoding
    def _n(self, v):          # NON_NEG
        return struct.pack(">q" if self.version == 5 else ">i", v)

    def _name(self, s):
        b = s.encode()
        return self._n(len(b)) + _pad4(b)

    def _att_list(self, atts):
        if not atts:
            return struct.pack(">i", 0) + self._n(0)
        out = struct.pack(">i", _TAG_ATT) + self._n(len(atts))
        for name, value in atts:
            if isinstance(value, (str, bytes)):
                b = value.encode() if isinstance(value, str) else value
                out += self._name(name) + struct.pack(">i", NC_CHAR) + self._n(len(b)) + _pad4(b)
            else:
                a = np.atleast_1d(np.asarray(value))
                t = _nc_type(a)
                out += (self._name(name) + struct.pack(">i", t) + self._n(a.size) +
                        _pad4(a.astype(_DTYPES[t]).tobytes()))
        return out

    def tobytes(self):
        def header(begins):
            h = b"CDF" + bytes([self.version]) + self._n(0)
            if self.dims:
                h += struct.pack(">i", _TAG_DIM) + self._n(len(self.dims))
                for name, size in self.dims:
                    h += self._name(name) + self._n(size)
            else:
                h += struct.pack(">i", 0) + self._n(0)
            h += self._att_list(self.gatts)
            if self.vars:
                h += struct.pack(">i", _TAG_VAR) + self._n(len(self.vars))
                for (name, ids, data, t, atts), begin in zip(self.vars, begins):
                    h += self._name(name) + self._n(len(ids))
                    for i in ids:
                        h += self._n(i)
                    h += self._att_list(list(atts.items())) + struct.pack(">i", t)
                    vsize = data.size * np.dtype(_DTYPES[t]).itemsize
                    h += self._n(vsize + (-vsize % 4)) + struct.pack(">q", begin)
            else:
                h += struct.pack(">i", 0) + self._n(0)
            return h

        blobs = [_pad4(d.astype(_DTYPES[t]).tobytes()) for _, _, d, t, _ in self.vars]
        hlen = len(header([0] * len(self.vars)))
        begins, pos = [], hlen
        for b in blobs:
            begins.append(pos)
            pos += len(b)
        return header(begins) + b"".join(blobs)

    def write(self, path):
        with open(path, "wb") as f:
            f.write(self.tobytes())


def read_cdf(path):
    """Reads a CDF-1/2/5 file without record variables: (dims dict, global attrs dict,
    variables dict name -> (array, attrs, dim names))."""
    buf = open(path, "rb").read()
    if buf[:8] == b"\x89HDF\r\n\x1a\n":
        raise ValueError(f"{path}: a netCDF-4 (HDF5) file; this reader speaks the classic formats -- convert "
                         f"with `nccopy -k cdf5 {path} out.nc` (the data model is the same)")
    if buf[:3] != b"CDF" or buf[3] not in (1, 2, 5):
        raise ValueError(f"{path}: not a classic-model NetCDF file (magic {buf[:4]!r})")
    v5 = buf[3] == 5
    pos = 4

    def n():
        nonlocal pos
        if v5:
            (x,) = struct.unpack_from(">q", buf, pos); pos += 8
        else:
            (x,) = struct.unpack_from(">i", buf, pos); pos += 4
        return x

    def i4():
        nonlocal pos
        (x,) = struct.unpack_from(">i", buf, pos); pos += 4
        return x

    def name():
        nonlocal pos
        k = n()
        s = buf[pos:pos + k].decode(); pos += k + (-k % 4)
        return s

    def att_list():
        nonlocal pos
        tag, cnt = i4(), n()
        out = {}
        if tag == 0:
            return out
        assert tag == _TAG_ATT
        for _ in range(cnt):
            nm, t, k = name(), i4(), n()
            size = k * np.dtype(_DTYPES[t]).itemsize
            raw = buf[pos:pos + size]; pos += size + (-size % 4)
            out[nm] = raw.decode() if t == NC_CHAR else np.frombuffer(raw, dtype=_DTYPES[t]).copy()
        return out

    if n() != 0:
        raise ValueError("record variables are not supported")
    dims = []
    tag, cnt = i4(), n()
    if tag == _TAG_DIM:
        for _ in range(cnt):
            nm = name(); dims.append((nm, n()))
    gatts = att_list()
    variables = {}
    tag, cnt = i4(), n()
    if tag == _TAG_VAR:
        for _ in range(cnt):
            nm = name()
            ids = [n() for _ in range(n())]
            atts = att_list()
            t = i4(); n()   # vsize
            if buf[3] == 1:
                begin = i4()
            else:
                (begin,) = struct.unpack_from(">q", buf, pos); pos += 8
            shape = tuple(dims[i][1] for i in ids)
            cnt_el = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(buf, dtype=_DTYPES[t], count=cnt_el, offset=begin).reshape(shape)
            if t == NC_CHAR:
                arr = arr.tobytes().rstrip(b"\x00").decode()
            else:
                arr = arr.astype(arr.dtype.newbyteorder("=")).copy()
            variables[nm] = (arr, atts, [dims[i][0] for i in ids])
    return dict(dims), gatts, variables


# --------------------------------------------------------------------- columns

def column_filename(prefix, ix, iy, index):
    """`<prefix>_<ix:03d>_<iy:03d>_<index:08d>.nc` (wrf_pmc_driver.F90:1386-1388,
    wrf_pmc_init.F90:1143-1145)."""
    return f"{prefix}_{ix:03d}_{iy:03d}_{index:08d}.nc"


def component_records(n_orig, least_ct, greatest_ct, comp_len=None, comp_source=None, comp_time=None):
    """aero_component_len / _start_ind / _particle_num / _source_num / _create_time
    (wrf_pmc_driver.F90:1421-1437) of n_total particles: the entries of the lists given
    (comp_len [n], comp_source [n][K] 0-based, comp_time [n][K]; None: no lists), then what the
    histogram n_orig [n_source][n] holds beyond them, in source order -- created at the particle's
    greatest create time, or at its least if it is the particle's first entry."""
    n_orig = np.asarray(n_orig, dtype=np.int64)
    n_source, nt = n_orig.shape
    least_ct = np.asarray(least_ct, dtype=np.float64); greatest_ct = np.asarray(greatest_ct, dtype=np.float64)
    if comp_len is None:
        comp_len = np.zeros(nt, np.int64); comp_source = np.zeros((nt, 0), np.int64); comp_time = np.zeros((nt, 0))
    comp_len = np.asarray(comp_len, dtype=np.int64)
    comp_source = np.asarray(comp_source, dtype=np.int64).reshape(nt, -1)
    comp_time = np.asarray(comp_time, dtype=np.float64).reshape(nt, -1)
    K = comp_source.shape[1]
    held = np.arange(K)[None, :] < comp_len[:, None]
    have = np.zeros((n_source, nt), np.int64)
    for s_ in range(n_source):
        have[s_] = ((comp_source == s_) & held).sum(1)
    extra = np.maximum(n_orig - have, 0)
    length = np.maximum(comp_len + extra.sum(0), 1)           # (a particle without a source: one entry)
    start = np.concatenate([[1], 1 + np.cumsum(length)[:-1]]) if nt else np.zeros(0, np.int64)
    ncomp = int(length.sum())
    part = np.repeat(np.arange(1, nt + 1), length)
    src = np.ones(ncomp, dtype=np.int64)
    tim = np.repeat(least_ct, length)
    for i in np.flatnonzero((comp_len > 0) | (length > 1) | (n_orig.sum(0) > 0)):
        at = int(start[i]) - 1
        k = int(comp_len[i])
        src[at:at + k] = comp_source[i, :k] + 1
        tim[at:at + k] = comp_time[i, :k]
        for s_ in range(n_source):
            c = int(extra[s_, i])
            src[at + k:at + k + c] = s_ + 1
            tim[at + k:at + k + c] = greatest_ct[i]
            if k == 0 and c > 0:
                tim[at] = least_ct[i]
            k += c
    return length, start, part, src, tim


def write_column(path, *, species_names, density, part_offset, vol, wclass, ids, least_ct,
                 greatest_ct, n_orig, weight_magnitude, next_id, time=0.0, del_t=0.0, index=0,
                 kappa=None, molec_weight=None, num_ions=None, source_names=None,
                 gas_names=(), gas_mixing_ratio=None, version=5, uuid="", cell_ix=1, cell_iy=1,
                 comp_len=None, comp_source=None, comp_time=None):
    """One column (nz levels) in the layout of output_column_to_file_new.

    part_offset [nz+1]; vol [n_spec][n_total] (m^3); wclass/ids/least_ct/greatest_ct [n_total];
    n_orig [n_source][n_total] or None; weight_magnitude [nz][n_class] (one weight group, flat
    weights: AERO_STATE_WEIGHT_FLAT_SPECIFIED, wrf_pmc_init.F90:252-253); comp_len / comp_source /
    comp_time: the particles' component lists (component_records)."""
    po = np.asarray(part_offset, dtype=np.int64)
    nz, nt = po.size - 1, int(po[-1])
    vol = np.asarray(vol, dtype=np.float64).reshape(len(species_names), nt)
    density = np.asarray(density, dtype=np.float64)
    wmag = np.asarray(weight_magnitude, dtype=np.float64).reshape(nz, -1)
    n_class = wmag.shape[1]
    wclass = np.asarray(wclass, dtype=np.int64)
    if n_orig is None or len(n_orig) == 0:
        n_orig = np.ones((1, nt), dtype=np.int64)
    n_orig = np.asarray(n_orig, dtype=np.int64).reshape(-1, nt)
    n_source = n_orig.shape[0]
    source_names = list(source_names or [f"source_{k + 1}" for k in range(n_source)])
    level = np.repeat(np.arange(nz), np.diff(po))
    comp_len, comp_start, comp_part, comp_src, comp_time = component_records(
        n_orig, least_ct, greatest_ct, comp_len, comp_source, comp_time)
    ncomp = int(comp_len.sum())

    w = CdfWriter(version)
    w.att("title", "PartMC version WRF-PartMC output file")
    w.att("source", "WRF-PartMC (pmcx device state)")
    w.att("UUID", uuid)
    w.att("Conventions", "CF-1.4")
    w.att("cell_ix", np.int32(cell_ix)); w.att("cell_iy", np.int32(cell_iy))
    d_spec = w.dim("aero_species", len(species_names))
    w.dim("aero_source", n_source)
    w.dim("number_levels", nz)
    w.dim("aero_particle", nt)
    w.dim("aero_components", ncomp)
    w.dim("aero_weight_group", 1)
    w.dim("aero_weight_class", n_class)
    if len(gas_names):
        w.dim("gas_species", len(gas_names))
    w.var("time", [], [float(time)], unit="s", description="time elapsed since simulation start")
    w.var("timestep", [], [float(del_t)], unit="s", description="current timestep size")
    w.var("timestep_index", [], np.array([index], dtype=np.int32))
    w.var("aero_species", ["aero_species"], np.arange(1, len(species_names) + 1, dtype=np.int32),
          names=",".join(species_names), description="dummy dimension variable (no useful value)")
    w.var("aero_source", ["aero_source"], np.arange(1, n_source + 1, dtype=np.int32),
          names=",".join(source_names))
    w.var("aero_density", ["aero_species"], density, unit="kg/m^3")
    if kappa is not None:
        w.var("aero_kappa", ["aero_species"], np.asarray(kappa, dtype=np.float64), unit="1")
    if molec_weight is not None:
        w.var("aero_molec_weight", ["aero_species"], np.asarray(molec_weight, dtype=np.float64), unit="kg/mol")
    if num_ions is not None:
        w.var("aero_num_ions", ["aero_species"], np.asarray(num_ions, dtype=np.int32))
    if len(gas_names):
        w.var("gas_species", ["gas_species"], np.arange(1, len(gas_names) + 1, dtype=np.int32),
              names=",".join(gas_names))
        gm = np.zeros((nz, len(gas_names))) if gas_mixing_ratio is None else np.asarray(gas_mixing_ratio)
        w.var("gas_mixing_ratio", ["gas_species", "number_levels"], gm.T, unit="ppb",
              long_name="mixing ratios of gas species")
    w.var("part_start_index", ["number_levels"], (po[:-1] + 1).astype(np.int32), unit="(1)",
          long_name="first particle index for each grid cell")
    w.var("n_parts", ["number_levels"], np.diff(po).astype(np.int32), unit="(1)",
          long_name="number of particles for each grid cell")
    w.var("aero_particle_mass", ["aero_species", "aero_particle"], vol * density[:, None], unit="kg",
          long_name="constituent masses of each aerosol particle")
    w.var("aero_component_len", ["aero_particle"], comp_len.astype(np.int32),
          long_name="number of aero_components for each aerosol particle")
    w.var("aero_component_start_ind", ["aero_particle"], comp_start.astype(np.int32),
          long_name="start index of aero_component for each aerosol particle")
    w.var("aero_component_particle_num", ["aero_components"], comp_part.astype(np.int32),
          long_name="associated aerosol particle number for each component")
    w.var("aero_component_source_num", ["aero_components"], comp_src.astype(np.int32),
          long_name="associated source number for each component")
    w.var("aero_component_create_time", ["aero_components"], comp_time,
          long_name="associated creation time for each component")
    w.var("aero_particle_weight_group", ["aero_particle"], np.ones(nt, dtype=np.int32),
          long_name="weight group number of each aerosol particle")
    w.var("aero_particle_weight_class", ["aero_particle"], (wclass + 1).astype(np.int32),
          long_name="weight class number of each aerosol particle")
    w.var("aero_water_hyst_leg", ["aero_particle"], np.zeros(nt, dtype=np.int32),
          long_name="leg of the water hysteresis curve leg of each aerosol particle")
    w.var("aero_num_conc", ["aero_particle"], wmag[level, wclass] if nt else np.zeros(0), unit="m^{-3}",
          long_name="number concentration for each particle")
    w.var("aero_id", ["aero_particle"], np.asarray(ids, dtype=np.int64),
          nc_type=NC_INT64 if version == 5 else NC_INT,
          long_name="unique ID number of each aerosol particle")
    w.var("aero_least_create_time", ["aero_particle"], np.asarray(least_ct, dtype=np.float64), unit="s",
          long_name="least creation time of each aerosol particle",
          description="least (earliest) creation time of any original constituent particles that "
                      "coagulated to form each particle, measured from the start of the simulation")
    w.var("aero_greatest_create_time", ["aero_particle"], np.asarray(greatest_ct, dtype=np.float64),
          unit="s", long_name="greatest creation time of each aerosol particle",
          description="greatest (latest) creation time of any original constituent particles that "
                      "coagulated to form each particle, measured from the start of the simulation")
    wshape = ["aero_weight_class", "aero_weight_group", "number_levels"]
    w.var("weight_type", wshape, np.full((n_class, 1, nz), 1, dtype=np.int32),
          description="type of each aerosol weighting function: 0 = invalid, 1 = none (w(D) = 1), "
                      "2 = power (w(D) = (D/D_0)^alpha), 3 = MFA (mass flow) (w(D) = (D/D_0)^(-3))")
    w.var("weight_magnitude", wshape, wmag.T.reshape(n_class, 1, nz), unit="m^{-3}",
          description="magnitude for each weighting function")
    w.var("weight_exponent", wshape, np.zeros((n_class, 1, nz)), unit="1",
          description="exponent alpha for the power weight_type, set to -3 for MFA, and zero otherwise")
    w.var("next_id", [], np.array([next_id], dtype=np.int64),
          nc_type=NC_INT64 if version == 5 else NC_INT, unit="1",
          description="next_id for particle creation")
    w.write(path)
    return path


def read_column(path):
    """The restart reader's view of a column file (init_read_in_restart,
    wrf_pmc_init.F90:1084-1286): arrays ready for `CellBlock.upload_cells` /
    `set_weights` / `set_next_id`."""
    dims, gatts, v = read_cdf(path)
    get = lambda k: v[k][0]
    n_parts = get("n_parts").astype(np.int64)
    start = get("part_start_index").astype(np.int64)
    po = np.concatenate([[0], np.cumsum(n_parts)])
    assert np.array_equal(start, po[:-1] + 1), "part_start_index is not the running sum of n_parts"
    nt = int(po[-1])
    density = get("aero_density")
    vol = get("aero_particle_mass") / density[:, None]          # :1232 vol = mass / density
    n_source = dims.get("aero_source", 1)
    n_orig = np.zeros((n_source, nt), dtype=np.int32)
    cpart, csrc = get("aero_component_particle_num"), get("aero_component_source_num")
    np.add.at(n_orig, (csrc - 1, cpart - 1), 1)
    # the lists themselves, padded to the longest (-1 / 0 past a particle's end): what
    # CellBlock.upload_components takes, cut to the block's n_comp_max by the caller
    clen = get("aero_component_len").astype(np.int64)
    cstart = get("aero_component_start_ind").astype(np.int64)
    assert np.array_equal(cstart, np.concatenate([[1], 1 + np.cumsum(clen)[:-1]])[:nt]), \
        "aero_component_start_ind is not the running sum of aero_component_len"
    assert np.array_equal(cpart, np.repeat(np.arange(1, nt + 1), clen)), "aero_component_particle_num"
    kmax = int(clen.max()) if nt else 0
    comp_source = np.full((nt, kmax), -1, dtype=np.int32); comp_time = np.zeros((nt, kmax))
    within = np.arange(int(clen.sum())) - np.repeat(cstart - 1, clen)
    comp_source[cpart - 1, within] = csrc - 1
    comp_time[cpart - 1, within] = get("aero_component_create_time")
    wmag = get("weight_magnitude")                               # [class][group][level]
    scalar = lambda k: np.asarray(get(k)).reshape(-1)[0]         # (no dimension, or one of length 1)
    return dict(
        comp_len=clen.astype(np.int32), comp_source=comp_source, comp_time=comp_time,
        species_names=v["aero_species"][1].get("names", "").split(","), density=density,
        part_offset=po, vol=vol, wclass=(get("aero_particle_weight_class") - 1).astype(np.int32),
        ids=get("aero_id").astype(np.int64), least_ct=get("aero_least_create_time"),
        greatest_ct=get("aero_greatest_create_time"), n_orig=n_orig,
        weight_magnitude=np.ascontiguousarray(wmag[:, 0, :].T), next_id=int(scalar("next_id")),
        time=float(scalar("time")), del_t=float(scalar("timestep")),
        index=int(scalar("timestep_index")), num_conc=get("aero_num_conc"),
        gas_mixing_ratio=(get("gas_mixing_ratio").T if "gas_mixing_ratio" in v else None))
