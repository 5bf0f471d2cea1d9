"""What the compiler made of the kernels whose speed hangs on a resource figure (hipcc's kernel-resource-usage remarks, kept by the build under
hand_tracking_samples_amd/build/*.usage.txt).  No GPU needed: hipcc cross-compiles."""
import pytest

from hand_tracking_samples_amd import build


@pytest.fixture(scope="module")
def usage():
    u = build.resource_usage()
    if not u:
        build.build(force=True, verbose=False)
        u = build.resource_usage()
    assert u, "no resource remarks: did the build run?"
    return u


def _one(usage, *parts):
    hits = [k for k in usage if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return usage[hits[0]]


def test_solver_builds(usage):
    """k_solve: no scratch memory in any build an update can launch; the small build is exactly 40 LDS allocation units (eight frames per CU) and fits two waves per
    SIMD, the one a 1024-frame batch takes fits four times into a CU's 160 KB."""
    small, only = _one(usage, "k_solveILi34ELi584ELi84ELi0E"), _one(usage, "k_solveILi66ELi1024ELi126ELi1024E")
    for k in usage:
        if "k_solve" in k and "ELb1E" not in k and "k_solveILi2ELi64E" not in k:      # ELb1E: the exact-order instantiation; <2, 64, ...>: the build that keeps every array in HBM (both tests only)
            assert usage[k]["ScratchSize"] == 0 and usage[k]["VGPRs Spill"] == 0, k
    assert small["LDS Size"] == 20480
    assert 4 * only["LDS Size"] <= 160 * 1024
    # round 5: the blocked two-body phases keep 64 coupling registers; eight frames per CU = two waves per SIMD = 256 registers with the accumulator file
    assert small["VGPRs"] + small["AGPRs"] <= 256 and small["Occupancy"] >= 2


def test_reset_kernel(usage):
    """k_reset: the build an update launches holds everything in registers (no private segment); the two-blocks-per-CU build stays within 256."""
    few, many = _one(usage, "k_resetILi1ELb0E"), _one(usage, "k_resetILi2ELb0E")      # Lb1 = the exact-order instantiation of the tests (ht_debug_solver_build 5)
    assert few["ScratchSize"] == 0
    assert many["VGPRs"] + many["AGPRs"] <= 256 and many["Occupancy"] >= 2


def test_closest_feature_and_cnn_kernels(usage):
    """Four blocks per CU for the closest-feature kernels (128 VGPRs at most, no scratch); the CNN's matrix kernels without scratch."""
    for name in ("k_cloud_rows", "k_fit_error"):
        k = _one(usage, name)
        assert k["ScratchSize"] == 0 and k["VGPRs"] <= 128, name
    for k in usage:
        if any(n in k for n in ("k_conv1", "k_conv2", "k_fcI", "k_fc144")):
            assert usage[k]["ScratchSize"] == 0, k


def test_cooperative_contact_kernel(usage):
    """k_contacts_coop (two waves per SIMD: 256 registers a wave): everything in registers, no private segment.  Until round 6 a run's two simplices kept p = a - b beside a
    and b (24 registers), and the kernel spilled 32 registers to 76 bytes of scratch per lane."""
    k = _one(usage, "k_contacts_coop")
    assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0 and k["VGPRs"] + k["AGPRs"] <= 256 and k["Occupancy"] >= 2
    assert _one(usage, "k_solve_prep")["ScratchSize"] == 0
