"""Thompson lookup-table caches in the reference's own file format (src/physics/mp_thompson.f90:2870-2887, 3037-3052,
3226-3240, 3291-3300, 3385-3392): Fortran UNFORMATTED SEQUENTIAL files `qr_acr_qg_mpt.dat`, `qr_acr_qs_mpt.dat`,
`freezeH2O_mpt.dat` in the working directory, one record per table (4-byte little-endian length, the REAL(8) table in
Fortran order, the length again).  A run of the reference that finds the files skips its 56 s table integration; the
files written here from the device tables are byte-identical to the ones the reference writes (tests/golden/
thompson_cache_sha256.json holds the digests of the reference's files)."""
import ctypes
import os
import struct
import numpy as np
from .capi import lib, check

FILES = {
    "qr_acr_qg_mpt.dat": ["tcg_racg", "tmr_racg", "tcr_gacr", "tmg_gacr", "tnr_racg", "tnr_gacr"],
    "qr_acr_qs_mpt.dat": ["tcs_racs1", "tmr_racs1", "tcs_racs2", "tmr_racs2", "tcr_sacr1", "tms_sacr1", "tcr_sacr2", "tms_sacr2",
                          "tnr_racs1", "tnr_racs2", "tnr_sacr1", "tnr_sacr2"],
    "freezeH2O_mpt.dat": ["tpi_qrfz", "tni_qrfz", "tpg_qrfz", "tnr_qrfz", "tpi_qcfz", "tni_qcfz"],
}


def device_table(domain, name):
    n = ctypes.c_size_t()
    check(lib().icar_hip_thompson_table(domain.ctx, name.encode(), None, ctypes.c_size_t(0), ctypes.byref(n)), "table size")
    out = np.empty(n.value, np.float64)
    check(lib().icar_hip_thompson_table(domain.ctx, name.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), None), "table")
    return out


def write_record(f, a):
    b = np.ascontiguousarray(a, "<f8").tobytes()
    f.write(struct.pack("<i", len(b))); f.write(b); f.write(struct.pack("<i", len(b)))


def read_records(path):
    out = []
    with open(path, "rb") as f:
        while True:
            h = f.read(4)
            if not h:
                break
            n = struct.unpack("<i", h)[0]
            a = np.frombuffer(f.read(n), "<f8").copy()
            if struct.unpack("<i", f.read(4))[0] != n:
                raise ValueError(f"{path}: record trailer does not match its header")
            out.append(a)
    return out


def write_caches(domain, directory="."):
    """After mp_init(options, domain): write the three cache files from the device tables."""
    for fname, tables in FILES.items():
        with open(os.path.join(directory, fname), "wb") as f:
            for t in tables:
                write_record(f, device_table(domain, t))


def read_caches(directory="."):
    """{table name: flat float64 array} from whichever of the three files exist."""
    out = {}
    for fname, tables in FILES.items():
        p = os.path.join(directory, fname)
        if os.path.exists(p):
            recs = read_records(p)
            if len(recs) != len(tables):
                raise ValueError(f"{p}: {len(recs)} records, expected {len(tables)}")
            out.update(dict(zip(tables, recs)))
    return out
