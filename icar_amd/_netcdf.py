"""What the NetCDF side of this package can and cannot interchange with a reference ICAR run.

The reference writes NetCDF-4 / HDF5 everywhere: output and restart files (`nf90_create(..., NF90_NETCDF4)`,
src/io/output_obj.f90:41-78), the linear-wind LUT cache (src/io/lt_lut_io.f90:58-108) and, through xarray, the ideal-case inputs
of helpers/genNetCDF.  This image has no HDF5 library, so everything here is NetCDF CLASSIC (CDF-1 / CDF-2 via
scipy.io.netcdf_file) with the reference's variable names, dimension names and orders:

    written here  -> read by the reference : yes.  nf90_open / nf90_inq_varid / nf90_get_var read classic files like NetCDF-4 ones.
    written by the reference -> read here  : only after `nccopy -k classic` (or `-k 64-bit-offset`); an HDF5 file is REJECTED
                                              with that advice, not parsed.
    limits of the classic format           : one record dimension, fixed-size variables < 4 GiB (a production LUT of 30 GB does
                                              not fit: write_LUT refuses it).

FORMAT_NOTE goes into the global attributes of every file this package writes."""
from .capi import IcarHipError

HDF5_SIGNATURE = b"\x89HDF\r\n\x1a\n"
FORMAT_NOTE = ("NetCDF classic written without an HDF5 library (icar_amd on MI355X); the reference ICAR writes NetCDF-4: its "
               "nf90_open reads this file as it is, a file the reference wrote needs `nccopy -k classic` before icar_amd reads it")


def open_classic(path, mode="r", **kw):
    """scipy.io.netcdf_file(path, mode), after a look at the signature when reading: NetCDF-4 / HDF5 is refused with advice."""
    from scipy.io import netcdf_file
    if mode == "r":
        with open(path, "rb") as f:
            head = f.read(8)
        if head == HDF5_SIGNATURE:
            raise IcarHipError(f"{path} is a NetCDF-4 / HDF5 file (what the reference ICAR and its xarray helpers write); this build "
                               "has no HDF5 library and reads NetCDF classic only: convert it with `nccopy -k classic` "
                               "(or `-k 64-bit-offset`), names and dimension orders stay the same")
        if head[:3] != b"CDF":
            raise IcarHipError(f"{path} is not a NetCDF classic file (signature {head[:4]!r})")
        kw.setdefault("mmap", False)
    return netcdf_file(path, mode, **kw)
