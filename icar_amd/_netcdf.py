"""What the NetCDF side of this package can and cannot interchange with a reference ICAR run.

Formats of the reference, file by file:
    output and restart files      NetCDF CLASSIC: `nf90_create(filename, NF90_CLOBBER, ...)`, src/io/output_obj.f90:54 -- the SAME format
                                  this package writes (CDF-1 via scipy.io.netcdf_file), with the reference's variable names, dimension
                                  names and orders and attributes (tests/golden/output_metadata.json is derived from
                                  src/io/default_output_metadata.f90 + output_obj.f90; tests/test_output_netcdf.py holds the writer to it).
                                  Either side reads the other's files as they are.
    linear-wind LUT cache         NetCDF-4 / HDF5: `nf90_create(..., NF90_NETCDF4)`, src/io/lt_lut_io.f90:323,403.  This image has no HDF5
                                  library: the cache written here is classic (the reference's nf90_open reads it); a cache the reference
                                  wrote needs `nccopy -k classic` (or `-k 64-bit-offset`) before it is read here -- an HDF5 file is
                                  REJECTED with that advice, not parsed.  Classic limits: fixed-size variables < 4 GiB (a production LUT
                                  of 30 GB does not fit: write_LUT refuses it).
    ideal-case / forcing inputs   whatever the tool that made them wrote; xarray (helpers/genNetCDF) defaults to NetCDF-4: same advice.

FORMAT_NOTE goes into the global attributes of every file this package writes."""
from .capi import IcarHipError

HDF5_SIGNATURE = b"\x89HDF\r\n\x1a\n"
FORMAT_NOTE = ("NetCDF classic (CDF-1), the format of the reference ICAR's own output and restart files (nf90_create NF90_CLOBBER); "
               "written by icar_amd on MI355X without an HDF5 library: NetCDF-4 inputs (the reference's linear-wind LUT cache, "
               "xarray-written files) need `nccopy -k classic` before icar_amd reads them")


def open_classic(path, mode="r", **kw):
    """scipy.io.netcdf_file(path, mode), after a look at the signature when reading: NetCDF-4 / HDF5 is refused with advice."""
    from scipy.io import netcdf_file
    if mode == "r":
        with open(path, "rb") as f:
            head = f.read(8)
        if head == HDF5_SIGNATURE:
            raise IcarHipError(f"{path} is a NetCDF-4 / HDF5 file (the reference's linear-wind LUT cache and xarray-written inputs are; its output and "
                               "restart files are classic); this build "
                               "has no HDF5 library and reads NetCDF classic only: convert it with `nccopy -k classic` "
                               "(or `-k 64-bit-offset`), names and dimension orders stay the same")
        if head[:3] != b"CDF":
            raise IcarHipError(f"{path} is not a NetCDF classic file (signature {head[:4]!r})")
        kw.setdefault("mmap", False)
    return netcdf_file(path, mode, **kw)
