"""Child of tests/test_gpu_bench_ranks.py::test_bench_falls_back_to_the_host_transport_when_rccl_init_fails: bench.py with the
RCCL communicator's initialisation reported as failed on this image."""
import os
import sys

if os.environ.get("ICAR_TEST_STRICT_TRANSPORT") != "1":
    os.environ["ICAR_ALLOW_HOST_STAGED"] = "1"  # without it HaloComm.attach raises (tests/test_gpu_bench_ranks.py checks both)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icar_amd.capi as capi  # noqa: E402

_check = capi.check


def check(rc, what=""):
    if what == "icar_hip_comm_init":
        raise RuntimeError("simulated: ncclCommInitRank failed on this image")
    return _check(rc, what)


capi.check = check
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
