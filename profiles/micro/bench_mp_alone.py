"""bench.py with NOTHING issued beside the interior microphysics (wind setup and w_real diagnostic after the join instead):
what the overlap with the Thompson launch is worth.  python profiles/micro/bench_mp_alone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icar_amd.time_step as t
from icar_amd.microphysics import mp


def mp_and_halo(domain, options, dt, overlap=True, prepare_advection=True, beside_interior=()):
    domain.aux_fork()
    mp(domain, options, dt, halo=1)
    domain.halo_send()
    domain.aux_begin()
    try:
        mp(domain, options, dt, subset=1)
    finally:
        domain.aux_end()
    domain.aux_join()
    for fn in beside_interior:
        fn()
    domain.halo_retrieve()


t.mp_and_halo = mp_and_halo
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "40", "--warmup", "5"]
bench.main()
