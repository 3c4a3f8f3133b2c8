#!/usr/bin/env python
"""Writes the ideal-case input files of tests/test_ideal_io.py / tests/test_gpu_ideal_run.py: init.nc and forcing.nc with the
variable / dimension names and the formulas of the reference's generator (tests/gen_ideal_test.py:21 ->
helpers/genNetCDF/Topography.py, Forcing.py; restated in icar_amd/ideal_io.py), in NetCDF classic.  The tests call write()
into a temporary directory (the files are a few hundred kB and fully determined by these arguments, so they are regenerated
rather than stored).  usage: python tests/golden/make_ideal_files.py <dir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# gen_ideal_test.py's non-Schaer settings, scaled down: hi-res grid 48 x 40, 20 levels; forcing 5 cells wider on every side
NX, NY, NZ, DX = 48, 40, 20, 1000.0
NT_LO, NZ_LO, DZ_LO = 4, 31, 500.0
HILL_HEIGHT, N_HILLS = 1500.0, 1
U_VAL, V_VAL, QV_VAL = 10.0, 3.0, 0.004


def write(directory):
    from icar_amd import ideal_io
    os.makedirs(directory, exist_ok=True)
    init, forcing = os.path.join(directory, "init.nc"), os.path.join(directory, "forcing.nc")
    ideal_io.write_init(init, NX, NY, DX, DX, hill_height=HILL_HEIGHT, n_hills=N_HILLS)
    ideal_io.write_forcing(forcing, NT_LO, NZ_LO, NX + 10, NY + 10, dz_value=DZ_LO, dx=DX, dy=DX, u_val=U_VAL, v_val=V_VAL, qv_val=QV_VAL)
    return init, forcing


if __name__ == "__main__":
    print(write(sys.argv[1] if len(sys.argv) > 1 else "."))
