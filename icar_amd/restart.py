"""restart_interface mirror (src/io/restart.f90): load every variable of the output dataset from a restart / output file
back into the domain.  SURVEY.md 8(f) row 2.  Host-side I/O at the boundary; the arrays go to the device through
domain.set (icar_hip_upload).  Files are the NetCDF-classic ones icar_amd.output writes (and ICAR's own classic files:
same names, (time, level, lat, lon) order)."""
import numpy as np
from .capi import IcarHipError
from .output import METADATA, MEMBER


def get_image_filename(image_number, initial_filename, restart_time):
    """restart.f90:83-100: <restart_file><image, 6 digits>_<YYYY-MM-DD_hh-mm-ss>.nc with the HOUR field forced to "00"
    (`file_name(n-10:n-9) = "00"`: output files start at midnight and hold frames_per_outfile records)."""
    name = f"{initial_filename}{image_number:06d}_{restart_time.strftime('%Y-%m-%d_%H-%M-%S')}.nc"
    n = len(name)
    return name[:n - 11] + "00" + name[n - 9:]


def read_restart_data(domain, dataset, filename, time_step):
    """restart.f90:22-81.  time_step is the 1-based record (restart_step_in_file).  3-D variables come back from the
    file's (level, lat, lon) to data_3d(i,k,j) like `reshape(data_3d, order=[1,3,2])` (:52); sizes must match the
    current decomposition or the run stops like restart_domain_error (:102-110)."""
    from ._netcdf import open_classic
    with open_classic(filename) as f:
        for n in dataset.variables:
            name, dims, _ = METADATA[n]
            if name not in f.variables:
                raise IcarHipError(f"Error reading restart variable: {name} (not in {filename})")
            v = f.variables[name]
            a = np.array(v[time_step - 1] if dims[0] == "time" else v[:])
            a = a.astype(a.dtype.newbyteorder("="))
            if a.ndim == 3:
                a = np.ascontiguousarray(a.transpose(1, 0, 2))                 # (level, lat, lon) -> (j, k, i)
            want = tuple(domain.shape(domain.fid(MEMBER[n])))
            if a.shape != want:
                raise IcarHipError(f"Error reading restart variable: {name}\n The domain of the restart file does not match the "
                                   f"current run (file {a.shape}, tile {want}); this can happen if you run a different "
                                   "number of parallel processes")
            domain.set(MEMBER[n], a)


def restart_model(domain, dataset, options):
    """restart.f90:11-20."""
    read_restart_data(domain, dataset, options.parameters.restart_file, options.parameters.restart_step_in_file)
