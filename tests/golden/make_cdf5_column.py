"""Assembles tests/golden/cdf5_column.hex: a WRF-PartMC column file with the full variable set of
output_column_to_file_new (interface/wrf_pmc_driver.F90:1478-1653 plus the time, gas-data and
aero-data variables written ahead of it, :1372-1379), laid out field by field from the NetCDF
classic-format grammar (netcdf-c docs/file_format_specifications.md, "The CDF-5 file format") and
annotated.  It shares no code with wrf_partmc_amd/ncio.py: tests/test_oracle.py holds the encoder
to these bytes and the decoder to these values.

    header   = magic numrecs dim_list gatt_list var_list
    magic    = 'C' 'D' 'F' 0x05            numrecs  = INT64 (0: no record dimension)
    dim_list = ABSENT | 0x0000000A nelems [dim ...]        dim = name dim_length
    att_list = ABSENT | 0x0000000C nelems [attr ...]       attr = name nc_type nelems [values ...] padding
    var_list = ABSENT | 0x0000000B nelems [var ...]
    var      = name nelems [dimid ...] vatt_list nc_type vsize begin
    name     = nelems namestring padding                   ABSENT = 0x00000000 INT64(0)
    nelems, dim_length, dimid, vsize, begin: INT64 in CDF-5; tags and nc_type: 4 bytes; data
    big-endian, each variable padded to a multiple of 4 bytes; vsize includes the padding.

The column: 2 levels; level 1 holds particles 1 and 2, level 2 particle 3; species SO4, BC; sources
traffic, sea; one weight class; gas o3.  Particle 2 coagulated twice: three constituents with
their own create times.  Run from the repo root: python tests/golden/make_cdf5_column.py"""
import os
import struct

NC_CHAR, NC_INT, NC_DOUBLE, NC_INT64 = 2, 4, 6, 10
SIZE = {NC_CHAR: 1, NC_INT: 4, NC_DOUBLE: 8, NC_INT64: 8}
PACK = {NC_INT: ">i", NC_DOUBLE: ">d", NC_INT64: ">q"}

out = []                       # (bytes, comment)


def put(b, what):
    out.append((b, what))


def i64(v, what):
    put(struct.pack(">q", v), what)


def i32(v, what):
    put(struct.pack(">i", v), what)


def name(s):
    i64(len(s), f"name: {len(s)} characters")
    put(s.encode() + b"\0" * (-len(s) % 4), f"'{s}'" + (" + padding" if len(s) % 4 else ""))


def values(t, vals, what):
    if t == NC_CHAR:
        b = vals.encode()
    else:
        b = b"".join(struct.pack(PACK[t], v) for v in vals)
    put(b + b"\0" * (-len(b) % 4), what)


def att_list(atts):
    if not atts:
        i32(0, "att_list ABSENT"); i64(0, "")
        return
    i32(0x0C, "NC_ATTRIBUTE"); i64(len(atts), f"{len(atts)} attributes")
    for nm, t, v in atts:
        name(nm)
        i32(t, "nc_type " + {2: "NC_CHAR", 4: "NC_INT", 6: "NC_DOUBLE"}[t])
        i64(len(v), "nelems")
        values(t, v, repr(v) if t == NC_CHAR else str(list(v)))


DIMS = [("aero_species", 2), ("aero_source", 2), ("number_levels", 2), ("aero_particle", 3),
        ("aero_components", 5), ("aero_weight_group", 1), ("aero_weight_class", 1), ("gas_species", 1)]
D = {n: k for k, (n, _) in enumerate(DIMS)}
LEN = dict(DIMS)
S = lambda s: (NC_CHAR, s)

# the particles (SI units): masses = vol * density
DENS = [1800.0, 1700.0]
VOL = [[1e-21, 2.5e-21, 4e-22], [0.0, 5e-22, 6e-22]]          # [species][particle]
MASS = [[VOL[s][p] * DENS[s] for p in range(3)] for s in range(2)]
CREATE_LEAST, CREATE_GREATEST = [0.0, 12.5, 300.0], [0.0, 86400.25, 300.0]
LONG_LEAST = ("least (earliest) creation time of any original constituent particles that coagulated to "
              "form each particle, measured from the start of the simulation")
LONG_GREATEST = LONG_LEAST.replace("least (earliest)", "greatest (latest)")
WT = ("type of each aerosol weighting function: 0 = invalid, 1 = none (w(D) = 1), 2 = power (w(D) = "
      "(D/D_0)^alpha), 3 = MFA (mass flow) (w(D) = (D/D_0)^(-3))")

# (name, dims slowest first, nc_type, values in file order, attributes)
VARS = [
    ("time", [], NC_DOUBLE, [7200.0], [("unit", *S("s")), ("description", *S("time elapsed since simulation start"))]),
    ("timestep", [], NC_DOUBLE, [60.0], [("unit", *S("s")), ("description", *S("current timestep size"))]),
    ("timestep_index", [], NC_INT, [120], []),
    ("aero_species", ["aero_species"], NC_INT, [1, 2],
     [("names", *S("SO4,BC")), ("description", *S("dummy dimension variable (no useful value)"))]),
    ("aero_source", ["aero_source"], NC_INT, [1, 2], [("names", *S("traffic,sea"))]),
    ("aero_density", ["aero_species"], NC_DOUBLE, DENS, [("unit", *S("kg/m^3"))]),
    ("aero_kappa", ["aero_species"], NC_DOUBLE, [0.65, 0.0], [("unit", *S("1"))]),
    ("aero_molec_weight", ["aero_species"], NC_DOUBLE, [0.096, 0.001], [("unit", *S("kg/mol"))]),
    ("gas_species", ["gas_species"], NC_INT, [1], [("names", *S("o3"))]),
    ("gas_mixing_ratio", ["gas_species", "number_levels"], NC_DOUBLE, [41.5, 38.25],
     [("unit", *S("ppb")), ("long_name", *S("mixing ratios of gas species"))]),
    ("part_start_index", ["number_levels"], NC_INT, [1, 3],
     [("unit", *S("(1)")), ("long_name", *S("first particle index for each grid cell"))]),
    ("n_parts", ["number_levels"], NC_INT, [2, 1],
     [("unit", *S("(1)")), ("long_name", *S("number of particles for each grid cell"))]),
    ("aero_particle_mass", ["aero_species", "aero_particle"], NC_DOUBLE, MASS[0] + MASS[1],
     [("unit", *S("kg")), ("long_name", *S("constituent masses of each aerosol particle"))]),
    ("aero_component_len", ["aero_particle"], NC_INT, [1, 3, 1],
     [("long_name", *S("number of aero_components for each aerosol particle"))]),
    ("aero_component_start_ind", ["aero_particle"], NC_INT, [1, 2, 5],
     [("long_name", *S("start index of aero_component for each aerosol particle"))]),
    ("aero_component_particle_num", ["aero_components"], NC_INT, [1, 2, 2, 2, 3],
     [("long_name", *S("associated aerosol particle number for each component"))]),
    ("aero_component_source_num", ["aero_components"], NC_INT, [1, 2, 1, 1, 2],
     [("long_name", *S("associated source number for each component"))]),
    ("aero_component_create_time", ["aero_components"], NC_DOUBLE, [0.0, 86400.25, 12.5, 3600.0, 300.0],
     [("long_name", *S("associated creation time for each component"))]),
    ("aero_particle_weight_group", ["aero_particle"], NC_INT, [1, 1, 1],
     [("long_name", *S("weight group number of each aerosol particle"))]),
    ("aero_particle_weight_class", ["aero_particle"], NC_INT, [1, 1, 1],
     [("long_name", *S("weight class number of each aerosol particle"))]),
    ("aero_water_hyst_leg", ["aero_particle"], NC_INT, [0, 0, 0],
     [("long_name", *S("leg of the water hysteresis curve leg of each aerosol particle"))]),
    ("aero_num_conc", ["aero_particle"], NC_DOUBLE, [2.5e7, 2.5e7, 1.25e7],
     [("unit", *S("m^{-3}")), ("long_name", *S("number concentration for each particle"))]),
    ("aero_id", ["aero_particle"], NC_INT64, [7, (3 << 32) + 11, (1 << 40) + 2],
     [("long_name", *S("unique ID number of each aerosol particle"))]),
    ("aero_least_create_time", ["aero_particle"], NC_DOUBLE, CREATE_LEAST,
     [("unit", *S("s")), ("long_name", *S("least creation time of each aerosol particle")),
      ("description", *S(LONG_LEAST))]),
    ("aero_greatest_create_time", ["aero_particle"], NC_DOUBLE, CREATE_GREATEST,
     [("unit", *S("s")), ("long_name", *S("greatest creation time of each aerosol particle")),
      ("description", *S(LONG_GREATEST))]),
    ("weight_type", ["aero_weight_class", "aero_weight_group", "number_levels"], NC_INT, [1, 1],
     [("description", *S(WT))]),
    ("weight_magnitude", ["aero_weight_class", "aero_weight_group", "number_levels"], NC_DOUBLE, [2.5e7, 1.25e7],
     [("unit", *S("m^{-3}")), ("description", *S("magnitude for each weighting function"))]),
    ("weight_exponent", ["aero_weight_class", "aero_weight_group", "number_levels"], NC_DOUBLE, [0.0, 0.0],
     [("unit", *S("1")), ("description", *S("exponent alpha for the power weight_type, set to -3 for MFA, and zero otherwise"))]),
    ("next_id", [], NC_INT64, [(1 << 40) + 3],
     [("unit", *S("1")), ("description", *S("next_id for particle creation"))]),
]
GATTS = [("title", *S("PartMC version WRF-PartMC output file")), ("source", *S("WRF-PartMC (pmcx device state)")),
         ("UUID", *S("c0ffee00-0000-4000-8000-000000000001")), ("Conventions", *S("CF-1.4")),
         ("cell_ix", NC_INT, [12]), ("cell_iy", NC_INT, [7])]


def assemble(begins):
    out.clear()
    put(b"CDF\x05", "magic: classic model, 64-bit data (CDF-5)")
    i64(0, "numrecs: no record dimension")
    i32(0x0A, "NC_DIMENSION"); i64(len(DIMS), f"{len(DIMS)} dimensions")
    for nm, ln in DIMS:
        name(nm); i64(ln, f"dim_length {ln}")
    att_list(GATTS)
    i32(0x0B, "NC_VARIABLE"); i64(len(VARS), f"{len(VARS)} variables")
    for (nm, dims, t, vals, atts), begin in zip(VARS, begins):
        name(nm)
        i64(len(dims), f"{len(dims)} dimensions" + (" (a scalar)" if not dims else ""))
        for d in dims:
            i64(D[d], f"dimid {D[d]} = {d}")
        att_list(atts)
        i32(t, "nc_type " + {4: "NC_INT", 6: "NC_DOUBLE", 10: "NC_INT64"}[t])
        n = 1
        for d in dims:
            n *= LEN[d]
        assert n == len(vals), nm
        vsize = n * SIZE[t]
        i64(vsize + (-vsize % 4), f"vsize: {n} x {SIZE[t]} bytes" + (" + padding" if vsize % 4 else ""))
        i64(begin, f"begin: file offset {begin}")
    header_len = sum(len(b) for b, _ in out)
    pos, nxt = header_len, []
    for nm, dims, t, vals, atts in VARS:
        nxt.append(pos)
        values(t, vals, f"data of {nm}: {vals}")
        pos += len(out[-1][0])
    return nxt


def main():
    begins = assemble([0] * len(VARS))
    assert assemble(begins) == begins          # (offsets have a fixed width: the header length does not move)
    here = os.path.dirname(os.path.abspath(__file__))
    total = sum(len(b) for b, _ in out)
    with open(os.path.join(here, "cdf5_column.hex"), "w") as f:
        f.write("# A WRF-PartMC column file (variable set of interface/wrf_pmc_driver.F90:1372-1379,1478-1653) in the\n"
                "# CDF-5 format, assembled field by field from the classic-format grammar by\n"
                "# tests/golden/make_cdf5_column.py (which shares no code with the encoder under test).\n"
                f"# {total} bytes; '#' starts a comment.\n")
        for b, what in out:
            for k in range(0, max(len(b), 1), 32):
                f.write(f"{b[k:k + 32].hex():<64s}  # {what if k == 0 else '...'}\n")
    print("wrote cdf5_column.hex:", total, "bytes")


if __name__ == "__main__":
    main()
