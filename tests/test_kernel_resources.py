"""Compiler luck as a test (CPU only: hipcc cross-compiles).  The two hot kernels run at a fixed number of
waves per SIMD with a register allocation that leaves nothing of their inner loops in scratch -- round 5 measured that ONE spilled
register inside the MPDATA march costs 4 %, and the one miscompile ever seen in the Thompson code needed SGPRs spilled into VGPR lanes.
None of that is visible in the source: a compiler update can change it silently.  These numbers are what the timings in
profiles/r06_* were measured with (profiles/r06_resources.json, written by profiles/collect_resources.py)."""
import pytest
from icar_amd import build as B


@pytest.fixture(scope="module")
def mpdata():
    return B.kernel_resources("mpdata.hip")


def test_mpdata_fused_no_scratch_in_any_variant(mpdata):
    fused = {k: v for k, v in mpdata.items() if k.startswith("k_mpdata_fused<")}
    assert len(fused) == 40                                             # KB 1..5 x FCT x PASS1 x EXACT
    for name, r in fused.items():
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0, (name, r)
        assert r["VGPRs"] <= 248, (name, r)                              # amdgpu_num_vgpr(124) of the unified file: room for a third, small wave


def test_mpdata_fused_metric_variant(mpdata):
    """the variant of the 512 x 512 x 40 metric grid and of the 8-GPU tile: 5 levels per thread, limiter, donor-cell pass inside,
    the waves hold exactly the column"""
    r = mpdata["k_mpdata_fused<5, true, true, true>"]
    assert r["Occupancy [waves/SIMD]"] == 2 and r["SGPRs Spill"] == 0 and r["VGPRs"] <= 248 and r["AGPRs"] == 0, r
    assert r["LDS Size [bytes/block]"] <= 160 * 1024, r


def test_thompson_pack_resources():
    res = B.kernel_resources("mp_thompson.hip")
    for n in ("k_thompson_pack<512>", "k_thompson_pack<1024>"):
        r = res[n]
        assert r["SGPRs Spill"] == 0, (n, r)                             # (SGPR spills into VGPR lanes on top of VGPR spills: the round-4 miscompile)
        assert r["Occupancy [waves/SIMD]"] == 4 and r["VGPRs"] <= 128, (n, r)
        assert r["VGPRs Spill"] <= 40 and r["ScratchSize [bytes/lane]"] <= 160, (n, r)


def test_side_stream_kernels_fit_beside_the_advection():
    """what runs on the second stream while k_mpdata_fused holds 2 x 248 of a SIMD's 512 registers must fit into the rest"""
    r = B.kernel_resources("capi.hip")["k_max_courant"]
    assert r["VGPRs"] <= 16 and r["ScratchSize [bytes/lane]"] == 0, r
