"""bench.py with nothing issued beside the advection (whole-field forcing after it, no CFL prefetch): what the second stream
costs / saves the MPDATA launch and the step.  python profiles/micro/bench_no_aux_overlap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icar_amd.time_step as t
t._FORCING_BESIDE_ADVECT = ()
_sub = t.substep
def substep(domain, options, dt, forced=None, diagnostics=True, enforce=False, prefetch_dt=True):
    return _sub(domain, options, dt, forced=forced, diagnostics=diagnostics, enforce=enforce, prefetch_dt=False)
t.substep = substep
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "40", "--warmup", "5"]
bench.main()
