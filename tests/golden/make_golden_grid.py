"""Generates tests/golden/grid_tiles.npz from the COMPILED REFERENCE (oracle/_ref/libicar_ref.so, grid_obj.f90):
every tile of several decompositions, incl. the src/tests/test_caf_other_image_grids.f90 scenario
(nx=1024, ny=1234, nz=13) and the staggered u/v grids (nx_extra / ny_extra).  Data only: the integer members."""
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref

CASES = [(1024, 1234, 13), (512, 512, 40), (100, 100, 30), (2000, 500, 20), (37, 411, 5)]
NIMAGES = [1, 2, 3, 4, 6, 7, 8, 12, 16, 36, 64]
rows = []
for (nx, ny, nz) in CASES:
    for n in NIMAGES:
        for ex in ((0, 0), (1, 0), (0, 1)):
            for img in range(1, n + 1):
                g = ref.grid(nx, ny, nz, n, img, *ex)
                rows.append([nx, ny, nz, n, img, ex[0], ex[1]] + [g[m] for m in ref.GRID_MEMBERS])
a = np.array(rows, np.int32)
np.savez_compressed(os.path.join(HERE, "grid_tiles.npz"), rows=a, members=np.array(ref.GRID_MEMBERS))
print("wrote grid_tiles.npz", a.shape)
