"""The product build against ITSELF as it stood at the end of round 5 (commit 43e6687): tests/golden/product_r05.npz holds, for the bench's 1024 frames (two updates
each, with and without always_take_cnn) and configs[4]'s 256 frames end to end, a hash of every frame's results and the results themselves for the first 48 frames
(tools/save_product_poses.py on the GPU box, condensed: the tool's docstring).  Round 6's changes to the solver move WHERE arithmetic runs (k_solve_prep beside the
contact kernel for the solve tables: tests/test_gpu_solve_tables.py), never what is computed: every pose, momentum and tracker flag must come out bit for bit the same."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys_path_tools = os.path.join(os.path.dirname(HERE), "tools")
FIX = np.load(os.path.join(HERE, "golden", "product_r05.npz"))


def _hash(a):
    a = np.ascontiguousarray(a)
    return np.array([int.from_bytes(hashlib.blake2b(a[i].tobytes(), digest_size=8).digest(), "little") for i in range(len(a))], np.uint64)


def _compare(got):
    bad = {}
    for k, v in got.items():
        differ = np.nonzero(_hash(v) != FIX["hash_" + k])[0]
        if len(differ):
            head = FIX["head_" + k]
            n = min(len(head), len(v))
            bad[k] = (len(differ), differ[:8].tolist(), float(np.abs(v[:n].astype(np.float64) - head[:n]).max()))
    assert not bad, "frames whose results differ from round 5's product build (count, first frames, largest move among the first 48): %s" % bad


def _load_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("save_product_poses", os.path.join(sys_path_tools, "save_product_poses.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("take_cnn", [0, 1])
def test_bench_frames_bit_for_bit_as_round_5(take_cnn):
    got = {"%s_take%d" % (k, take_cnn): v for k, v in _load_tool().run1024(take_cnn).items()}
    _compare(got)


def test_config5_frames_bit_for_bit_as_round_5():
    _compare(_load_tool().run_config5())


@pytest.mark.parametrize("build", [1, 3])
def test_other_solver_builds_return_the_same_bits(build):
    """the builds of k_solve differ in where a frame's arrays live (LDS or its HBM slot), never in arithmetic: the small build (batches above 1024 frames) and the mid build
    (larger models) on the 1024 frames"""
    got = {"%s_take0" % k: v for k, v in _load_tool().run1024(0, build).items()}
    _compare(got)
