"""Several trackers of one process driven from several host threads at once (one context per thread: a HandTracker object is not shared in the reference either, its
update() owns the object for the call).  The library's process-wide state is a handful of once-only set-ups (kernel attributes, the environment read-outs) behind a
mutex or a C++11 static; everything else hangs off the context.  Four threads x one context x a twelve-update stream each, all at once, must give every thread exactly the
poses its stream gives when it runs alone."""
import os
import threading

import numpy as np
import pytest

import oracle_lib as ol
from hand_tracking_samples_amd import weights as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FR = np.load(os.path.join(os.path.dirname(HERE), "bench_data", "frames1024.npz"))
N = len(FR["depth"])


def _stream(weights, tid, T, K, out, barrier=None):
    from hand_tracking_samples_amd import native
    depth = FR["depth"].reshape(N, -1)
    idx = [(7 * np.arange(T) + 131 * tid + k) % N for k in range(K)]
    ctx = native.Context(ol.MODEL, T)
    try:
        ctx.load_weights(weights)
        ctx.set_params(microforce=3.0, mainthreadpasses=3)
        ctx.tracker_reset(FR["startpose"][idx[0]])
        if barrier is not None:
            barrier.wait()      # all threads enter their first update together
        res = []
        for k in range(K):
            res.append(ctx.update_sync(depth[idx[k]], FR["cam"][idx[k]]))
        out[tid] = (np.stack(res), ctx.get_state(0, T), ctx.get_state(1, T), ctx.capacity_events())
    except BaseException as e:      # a thread's failure must reach the test
        out[tid] = e
    finally:
        ctx.close()


def test_four_threads_four_contexts_at_once():
    weights = W.make_cnnb()
    NT, T, K = 4, 96, 12
    alone = {}
    for tid in range(NT):
        _stream(weights, tid, T, K, alone)
        assert not isinstance(alone[tid], BaseException), alone[tid]
    together = {}
    barrier = threading.Barrier(NT)
    threads = [threading.Thread(target=_stream, args=(weights, tid, T, K, together, barrier)) for tid in range(NT)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "a thread did not come back"
    for tid in range(NT):
        assert not isinstance(together[tid], BaseException), together[tid]
        for a, b in zip(alone[tid][:3], together[tid][:3]):
            assert np.array_equal(a, b), "thread %d: results differ from the same stream run alone" % tid
        assert together[tid][3] == (0, 0, 0)
    print("four threads, a context each, twelve updates of %d trackers at once: every thread's poses and states equal its stream run alone" % T)
