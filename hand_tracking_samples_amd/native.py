"""ctypes binding of libht_mi355x.so (the C-ABI declared in include/ht_mi355x.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if it cannot be loaded, or a context cannot be
created on a gfx950 device, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
HT_OK = 0
CNN_IN, CNN_OUT, CNNB_COUNT, POSE, STATE, CAM, ANALYSIS = 4096, 2304, 9458400, 7, 13, 12, 84
MAXPTS, ROW, CONTACT = 4096, 16, 12      # HT_MAX_POINTS

# every symbol include/ht_mi355x.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "ht_create", "ht_destroy", "ht_model_bake", "ht_last_error", "ht_get_params", "ht_set_params", "ht_model_info", "ht_config_read", "ht_scale",
    "ht_cnn_load_weights", "ht_cnn_eval", "ht_cnn_eval_dev", "ht_cnn_load_weights_sized", "ht_cnn_eval_sized", "ht_cnn_eval_sized_dev", "ht_cnn_train", "ht_cnn_get_weights", "ht_expected_cnn", "ht_expected_cnn_full",
    "ht_model_open", "ht_model_close", "ht_model_error", "ht_model_counts", "ht_model_body", "ht_model_body_mesh", "ht_model_body_sdmesh", "ht_model_hitcheck",
    "ht_tracker_reset", "ht_get_state", "ht_set_state", "ht_get_tracker_flags", "ht_set_tracker_flags", "ht_update_sync", "ht_update_dev", "ht_update_frames_sync", "ht_update_frames_dev", "ht_update_direct_sync", "ht_update_direct_dev", "ht_update_cnn_model_sync", "ht_get_cnn_results", "ht_get_cnn_layers", "ht_frames_overflow", "ht_reserve_points", "ht_point_capacity", "ht_capacity_events", "ht_segment_vr", "ht_segment_vr_dev", "ht_slowfit", "ht_set_points", "ht_fit_rows", "ht_physics_update",
    "ht_stage_prepare", "ht_stage_decode", "ht_stage_fit_error", "ht_stage_cloud_rows", "ht_stage_contacts", "ht_stage_fit",
    "ht_stage_multistep", "ht_stage_multistep_range", "ht_stage_scratch_unibody", "ht_stage_chamber", "ht_profile_enable", "ht_profile_read", "ht_debug_solve_stats", "ht_debug_contact_stats", "ht_debug_solver_build", "ht_debug_reset_flags", "ht_debug_reset_organisation", "ht_update_passes_sync", "ht_job_start", "ht_job_poll", "ht_job_wait", "ht_job_collect", "ht_debug_contact_kernel", "ht_contact_capacity", "ht_debug_solve_tables", "ht_debug_solve_tables_header",
    "ht_comm_available", "ht_comm_unique_id", "ht_comm_init", "ht_comm_info", "ht_gather_poses_dev", "ht_gather_wait", "ht_gather_wait_host", "ht_comm_destroy",
)


class Params(C.Structure):
    _fields_ = [("full_reset_on_error", C.c_float), ("angles_only", C.c_int), ("always_take_cnn", C.c_int), ("drangey", C.c_float), ("boundary_planes", C.c_int),
                ("microforce", C.c_float), ("cloudforce_max_point", C.c_float), ("cloudforce_max_sum", C.c_float), ("mainthreadpasses", C.c_int),
                ("subsample_fraction", C.c_int), ("min_point_num", C.c_int), ("accum_error_threshold", C.c_float), ("min_cray_prob", C.c_float),
                ("steps", C.c_int), ("steps_keypoints", C.c_int), ("steps_keyangles", C.c_int), ("steps_palmangle", C.c_int), ("steps_cloudstart", C.c_int), ("steps_unibody", C.c_int),
                ("physics_iterations", C.c_int), ("physics_iterations_post", C.c_int), ("physics_use_collision", C.c_int),
                ("physics_weak_force", C.c_float), ("bone_sum_error_scale", C.c_float), ("unibody_force", C.c_float),
                ("subsample_voxel", C.c_int), ("subsample_size", C.c_float)]


class HTError(RuntimeError):
    pass


_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building first if needed) the shared library; raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and _build.stale():
        _build.build(verbose=False)
    if not os.path.exists(_build.LIB):
        raise HTError("libht_mi355x.so is missing and could not be built; the MI355X path has no fallback")
    L = C.CDLL(_build.LIB)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    u16p = C.POINTER(C.c_uint16)
    L.ht_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.ht_destroy.argtypes = [vp]
    L.ht_last_error.argtypes = [vp]; L.ht_last_error.restype = C.c_char_p
    L.ht_get_params.argtypes = [vp, C.POINTER(Params)]; L.ht_set_params.argtypes = [vp, C.POINTER(Params)]
    L.ht_model_info.argtypes = [vp, ip, ip, ip]
    L.ht_cnn_load_weights.argtypes = [vp, fp, C.c_size_t]
    L.ht_cnn_eval.argtypes = [vp, fp, fp, C.c_int]
    L.ht_cnn_eval_dev.argtypes = [vp, vp, vp, C.c_int, vp]
    L.ht_cnn_load_weights_sized.argtypes = [vp, C.c_int, fp, C.c_size_t]
    L.ht_cnn_eval_sized.argtypes = [vp, C.c_int, fp, fp, C.c_int]
    L.ht_cnn_eval_sized_dev.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.ht_tracker_reset.argtypes = [vp, C.c_int, C.c_int, fp]
    L.ht_get_state.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp]; L.ht_set_state.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp]
    L.ht_get_tracker_flags.argtypes = [vp, C.c_int, C.c_int, fp, ip]
    L.ht_set_tracker_flags.argtypes = [vp, C.c_int, C.c_int, fp, ip]
    L.ht_update_sync.argtypes = [vp, u16p, fp, C.c_int, fp, fp]
    L.ht_update_dev.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
    L.ht_update_frames_sync.argtypes = [vp, u16p, fp, C.c_int, C.c_int, C.c_float, C.c_int, fp, fp]
    L.ht_update_frames_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, vp]
    L.ht_update_direct_sync.argtypes = [vp, u16p, fp, C.c_int, C.c_int, fp, fp]
    L.ht_update_direct_dev.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
    L.ht_frames_overflow.argtypes = [vp, ip]
    L.ht_reserve_points.argtypes = [vp, C.c_int]
    L.ht_point_capacity.argtypes = [vp, ip]
    L.ht_update_cnn_model_sync.argtypes = [vp, u16p, fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, fp, ip, fp]
    L.ht_get_cnn_results.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp]
    L.ht_capacity_events.argtypes = [vp, ip, ip, ip]
    L.ht_stage_prepare.argtypes = [vp, u16p, fp, C.c_int, fp, fp, ip]
    L.ht_stage_decode.argtypes = [vp, fp, fp, C.c_int, fp]
    L.ht_stage_fit_error.argtypes = [vp, C.c_int, C.c_int, fp]
    L.ht_stage_cloud_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip]
    L.ht_stage_contacts.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, ip]
    L.ht_stage_fit.argtypes = [vp, C.c_int]
    L.ht_stage_multistep.argtypes = [vp, fp, C.c_int]
    L.ht_stage_multistep_range.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int]
    L.ht_stage_scratch_unibody.argtypes = [vp, fp, C.c_int, C.c_int]
    L.ht_profile_enable.argtypes = [vp, C.c_int]
    L.ht_scale.argtypes = [vp, C.c_float]
    L.ht_cnn_train.argtypes = [vp, fp, fp, C.c_int, C.c_float, fp]
    L.ht_cnn_get_weights.argtypes = [vp, fp, C.c_size_t]
    L.ht_expected_cnn.argtypes = [fp, fp, fp]
    L.ht_set_points.argtypes = [vp, C.c_int, fp, C.c_int, ip]
    L.ht_slowfit.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, C.c_int]
    L.ht_get_cnn_layers.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, fp]
    L.ht_stage_chamber.argtypes = [vp, C.c_int, C.c_int, fp, ip]
    L.ht_fit_rows.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int, ip, fp, C.c_int, ip, fp, C.c_int, ip, C.c_float]
    L.ht_physics_update.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int, ip, fp, C.c_int, ip]
    L.ht_segment_vr.argtypes = [vp, C.POINTER(C.c_uint16), fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint16), fp]
    L.ht_segment_vr_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp, vp, vp]
    L.ht_profile_read.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_int, fp, ip, ip]
    L.ht_debug_solve_stats.argtypes = [vp, C.c_int, fp, C.c_int]
    L.ht_debug_solver_build.argtypes = [vp, C.c_int]
    L.ht_debug_reset_organisation.argtypes = [vp, ip]
    L.ht_debug_reset_flags.argtypes = [vp, ip, C.c_int]
    L.ht_debug_contact_kernel.argtypes = [vp, C.c_int]
    L.ht_debug_solve_tables.argtypes = [vp, C.c_int]
    L.ht_contact_capacity.argtypes = [vp, ip, ip, ip, ip]
    L.ht_debug_solve_tables_header.argtypes = [vp, C.c_int, ip]
    L.ht_comm_unique_id.argtypes = [vp]
    L.ht_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ht_comm_info.argtypes = [vp, ip, ip]
    L.ht_gather_poses_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    L.ht_gather_wait.argtypes = [vp, C.c_int, vp]
    L.ht_gather_wait_host.argtypes = [vp, C.c_int]
    L.ht_comm_available.argtypes = []
    L.ht_comm_destroy.argtypes = [vp]
    L.ht_debug_contact_stats.argtypes = [vp, C.c_int, fp, C.c_int]
    for name in SYMBOLS:
        if name not in ("ht_last_error", "ht_model_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def model_bake(json_path, out_path, hand_tweaks=True):
    """Host-only model build (PhysModel ctor + LoadHandModel, physmodel.h:444-475, handtrack.h:347-366) -> baked container."""
    r = load().ht_model_bake(str(json_path).encode(), str(out_path).encode(), 1 if hand_tweaks else 0)
    if r != 0:
        raise HTError("ht_model_bake(%s) failed with status %d" % (json_path, r))
    return out_path


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Context:
    """One (GPU, stream) context = the batched counterpart of a HandTracker object (handtrack.h:513-846)."""

    def __init__(self, model_path, max_batch, device=0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.ht_create(None if model_path is None else os.fsencode(model_path), int(max_batch), int(device), C.byref(self.h))      # None: a context that serves the net only (ht_mi355x.h)
        if rc != HT_OK:
            msg = self.L.ht_last_error(self.h).decode() if self.h else "ht_create failed"
            if self.h:
                self.L.ht_destroy(self.h)
                self.h = C.c_void_p()
            raise HTError("ht_create: %s (status %d)" % (msg, rc))
        nb, nj, mb = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.ht_model_info(self.h, C.byref(nb), C.byref(nj), C.byref(mb)))
        self.nb, self.nj, self.max_batch = nb.value, nj.value, mb.value

    def close(self):
        if self.h:
            self.L.ht_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != HT_OK:
            raise HTError("%s (status %d)" % (self.L.ht_last_error(self.h).decode(), rc))

    # -- parameters
    @property
    def params(self):
        p = Params()
        self._chk(self.L.ht_get_params(self.h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.params
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self._chk(self.L.ht_set_params(self.h, C.byref(p)))

    # -- CNN
    def load_weights(self, w):
        w = _c(w, np.float32)
        self._chk(self.L.ht_cnn_load_weights(self.h, _f(w), w.size))

    def cnn_eval(self, x):
        x = _c(x, np.float32).reshape(-1, CNN_IN)
        out = np.empty((x.shape[0], CNN_OUT), np.float32)
        self._chk(self.L.ht_cnn_eval(self.h, _f(x), _f(out), x.shape[0]))
        return out

    def cnn_eval_dev(self, d_in, d_out, B, stream):
        self._chk(self.L.ht_cnn_eval_dev(self.h, d_in, d_out, B, stream))

    # -- the same layers on a 128x128 input (BASELINE configs[4])
    def load_weights128(self, w):
        w = _c(w, np.float32)
        self._chk(self.L.ht_cnn_load_weights_sized(self.h, 128, _f(w), w.size))

    def cnn128_eval(self, x):
        x = _c(x, np.float32).reshape(-1, 128 * 128)
        out = np.empty((x.shape[0], CNN_OUT), np.float32)
        self._chk(self.L.ht_cnn_eval_sized(self.h, 128, _f(x), _f(out), x.shape[0]))
        return out

    def cnn128_eval_dev(self, d_in, d_out, B, stream):
        self._chk(self.L.ht_cnn_eval_sized_dev(self.h, 128, d_in, d_out, B, stream))

    # -- tracker
    def tracker_reset(self, poses, first=0):
        poses = _c(poses, np.float32).reshape(-1, self.nb, POSE)
        self._chk(self.L.ht_tracker_reset(self.h, first, poses.shape[0], _f(poses)))

    def get_state(self, which, n, first=0):
        s = np.empty((n, self.nb, STATE), np.float32)
        self._chk(self.L.ht_get_state(self.h, which, first, n, _f(s)))
        return s

    def set_state(self, which, state, first=0):
        state = _c(state, np.float32).reshape(-1, self.nb, STATE)
        self._chk(self.L.ht_set_state(self.h, which, first, state.shape[0], _f(state)))

    def tracker_flags(self, n, first=0):
        e = np.empty(n, np.float32); i = np.empty(n, np.int32)
        self._chk(self.L.ht_get_tracker_flags(self.h, first, n, _f(e), _i(i)))
        return e, i

    def set_tracker_flags(self, prev_frame_error, initializing, first=0):
        e = _c(prev_frame_error, np.float32); i = _c(initializing, np.int32)
        self._chk(self.L.ht_set_tracker_flags(self.h, first, e.size, _f(e), _i(i)))

    def update_sync(self, depth, cams, want_cnn=False):
        depth = _c(depth, np.uint16).reshape(-1, 4096)
        cams = _c(cams, np.float32).reshape(-1, CAM)
        B = depth.shape[0]
        poses = np.empty((B, self.nb, POSE), np.float32)
        cnn = np.empty((B, CNN_OUT), np.float32) if want_cnn else None
        self._chk(self.L.ht_update_sync(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), B, _f(poses), _f(cnn) if want_cnn else None))
        return (poses, cnn) if want_cnn else poses

    def update_dev(self, d_depth, d_cams, d_start, B, d_poses_out, stream):
        self._chk(self.L.ht_update_dev(self.h, d_depth, d_cams, d_start, B, d_poses_out, stream))

    def update_frames_sync(self, depth, cams, segment_scale=0.17, want_cnn=False):
        """HandTracker::update on frames of any size (handtrack.h:693-785): depth u16[B,h,w], cams [B,12] = the frames' cameras."""
        depth = _c(depth, np.uint16); B, h, w = depth.shape
        cams = _c(cams, np.float32).reshape(B, CAM)
        poses = np.empty((B, self.nb, POSE), np.float32)
        cnn = np.empty((B, CNN_OUT), np.float32) if want_cnn else None
        self._chk(self.L.ht_update_frames_sync(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), w, h, float(segment_scale), B, _f(poses), _f(cnn) if want_cnn else None))
        return (poses, cnn) if want_cnn else poses

    def update_frames_dev(self, d_depth, d_cams, w, h, segment_scale, d_start, B, d_poses_out, stream):
        self._chk(self.L.ht_update_frames_dev(self.h, d_depth, d_cams, int(w), int(h), float(segment_scale), d_start, B, d_poses_out, stream))

    def update_direct_sync(self, depth, cams, side=128, want_cnn=False):
        """BASELINE configs[4] end to end: depth u16[B,side,side] frames that are their own segment, the net of that input size (ht_update_direct_sync)."""
        depth = _c(depth, np.uint16).reshape(-1, side * side)
        B = depth.shape[0]
        cams = _c(cams, np.float32).reshape(B, CAM)
        poses = np.empty((B, self.nb, POSE), np.float32)
        cnn = np.empty((B, CNN_OUT), np.float32) if want_cnn else None
        self._chk(self.L.ht_update_direct_sync(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), int(side), B, _f(poses), _f(cnn) if want_cnn else None))
        return (poses, cnn) if want_cnn else (poses, None)

    def update_direct_dev(self, d_depth, d_cams, side, d_start, B, d_poses_out, stream):
        self._chk(self.L.ht_update_direct_dev(self.h, d_depth, d_cams, int(side), d_start, B, d_poses_out, stream))

    def update_cnn_model_sync(self, depth, cams, kickstart=False, segment_scale=0.17):
        """HandTracker::update_cnn_model (handtrack.h:734-741) / kickstart (:743-746): depth u16[B,h,w] -> (othermodel poses [B,nb,7], accepted [B])."""
        depth = _c(depth, np.uint16)
        if depth.ndim == 2:
            depth = depth.reshape(-1, 64, 64)
        B, h, w = depth.shape
        cams = _c(cams, np.float32).reshape(B, CAM)
        poses = np.empty((B, self.nb, POSE), np.float32); acc = np.empty(B, np.int32)
        self._chk(self.L.ht_update_cnn_model_sync(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), w, h, float(segment_scale), B, int(bool(kickstart)), _f(poses), _i(acc), None))
        return poses, acc

    def cnn_results(self, n, first=0):
        """(cnn_input [n,4096], cnn_output [n,2304], analysis [n,84]) of the latest update of those slots (HandTracker::cnn_input / cnn_output / cnn_output_analysis)"""
        a = np.empty((n, CNN_IN), np.float32); b = np.empty((n, CNN_OUT), np.float32); c = np.empty((n, ANALYSIS), np.float32)
        self._chk(self.L.ht_get_cnn_results(self.h, first, n, _f(a), _f(b), _f(c)))
        return a, b, c

    def capacity_events(self):
        """(expanding-polytope runs cut short, contacts dropped, solves with angular rows over capacity) since the context was created"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.L.ht_capacity_events(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def cnn_layers(self, n, first=0):
        """(act1 [n,3600], act2 [n,2304], act3 [n,2048], logits [n,2304]) of the latest CNN evaluation (ht_get_cnn_layers)."""
        a1 = np.zeros((n, 3600), np.float32); a2 = np.zeros((n, 2304), np.float32); a3 = np.zeros((n, 2048), np.float32); lg = np.zeros((n, 2304), np.float32)
        self._chk(self.L.ht_get_cnn_layers(self.h, int(first), int(n), _f(a1), _f(a2), _f(a3), _f(lg)))
        return a1, a2, a3, lg

    def stage_chamber(self, which, B):
        """cloud_chamber rows [B, 5*nb, 16] and their counts (ht_stage_chamber)."""
        rows = np.zeros((B, 5 * self.nb, 16), np.float32); n = np.zeros(B, np.int32)
        self._chk(self.L.ht_stage_chamber(self.h, int(which), int(B), _f(rows), _i(n)))
        return rows, n

    @staticmethod
    def _ragged(rows, width):
        """list of (n_i, width) arrays -> ([B, cap, width] float32, counts int32, cap)"""
        cap = max(1, max((len(r) for r in rows), default=1))
        buf = np.zeros((len(rows), cap, width), np.float32)
        for i, r in enumerate(rows):
            if len(r):
                buf[i, :len(r)] = np.asarray(r, np.float32).reshape(-1, width)
        return buf, np.array([len(r) for r in rows], np.int32), cap

    def fit_rows(self, which, clouds, linears, angulars, microforce=1.0):
        """PhysModel::FitPointCloud(points, linears, angulars, microforce) (physmodel.h:345-356) on model `which` of slots [0, len(clouds)):
        per slot a point cloud (n,3), the caller's linear rows (n,16) and angular rows (n,8)."""
        B = len(clouds)
        pb, pn, pc = self._ragged(clouds, 3); lb, ln, lc = self._ragged(linears, 16); ab, an, ac = self._ragged(angulars, 8)
        self._chk(self.L.ht_fit_rows(self.h, int(which), B, _f(pb), pc, _i(pn), _f(lb), lc, _i(ln), _f(ab), ac, _i(an), float(microforce)))

    def physics_update(self, which, linears, angulars):
        """PhysicsUpdate(rigidbodies, Linears, Angulars) (physics.h:543-587) on model `which`: per slot the caller's rows (n,16) / (n,8) are all there is."""
        B = len(linears)
        lb, ln, lc = self._ragged(linears, 16); ab, an, ac = self._ragged(angulars, 8)
        self._chk(self.L.ht_physics_update(self.h, int(which), B, _f(lb), lc, _i(ln), _f(ab), ac, _i(an)))

    def set_points(self, clouds):
        """Caller-supplied clouds for the stage calls (ht_set_points): one (n_i, 3) array per slot."""
        cap = max(1, max(len(c) for c in clouds))
        buf = np.zeros((len(clouds), cap, 3), np.float32)
        for i, c in enumerate(clouds):
            buf[i, :len(c)] = np.asarray(c, np.float32).reshape(-1, 3)
        n = np.array([len(c) for c in clouds], np.int32)
        self._chk(self.L.ht_set_points(self.h, len(clouds), _f(buf), cap, _i(n)))

    def point_capacity(self):
        n = C.c_int(0)
        self._chk(self.L.ht_point_capacity(self.h, C.byref(n)))
        return n.value

    def reserve_points(self, points):
        self._chk(self.L.ht_reserve_points(self.h, int(points)))

    def frames_overflow(self):
        n = C.c_int(0)
        self._chk(self.L.ht_frames_overflow(self.h, C.byref(n)))
        return n.value

    # -- stages
    def stage_prepare(self, depth, cams):
        depth = _c(depth, np.uint16).reshape(-1, 4096)
        cams = _c(cams, np.float32).reshape(-1, CAM)
        B = depth.shape[0]
        cnn_in = np.empty((B, CNN_IN), np.float32); pts = np.empty((B, MAXPTS, 4), np.float32); n = np.empty(B, np.int32)
        self._chk(self.L.ht_stage_prepare(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), B, _f(cnn_in), _f(pts), _i(n)))
        return cnn_in, pts, n

    def stage_decode(self, cnn_out, cams):
        cnn_out = _c(cnn_out, np.float32).reshape(-1, CNN_OUT)
        cams = _c(cams, np.float32).reshape(-1, CAM)
        an = np.empty((cnn_out.shape[0], ANALYSIS), np.float32)
        self._chk(self.L.ht_stage_decode(self.h, _f(cnn_out), _f(cams), cnn_out.shape[0], _f(an)))
        return an

    def stage_fit_error(self, which, B):
        e = np.empty(B, np.float32)
        self._chk(self.L.ht_stage_fit_error(self.h, which, B, _f(e)))
        return e

    def stage_cloud_rows(self, which, stride, use_cam_origin, B):
        rows = np.empty((B, MAXPTS, ROW), np.float32); n = np.empty(B, np.int32)
        self._chk(self.L.ht_stage_cloud_rows(self.h, which, stride, int(use_cam_origin), B, _f(rows), _i(n)))
        return rows, n

    def stage_contacts(self, which, B, cap=192):
        c = np.empty((B, cap, CONTACT), np.float32); n = np.empty(B, np.int32)
        self._chk(self.L.ht_stage_contacts(self.h, which, B, cap, _f(c), _i(n)))
        return c, n

    def stage_fit(self, B):
        self._chk(self.L.ht_stage_fit(self.h, B))

    def stage_multistep(self, analysis, B):
        analysis = _c(analysis, np.float32).reshape(B, ANALYSIS)
        self._chk(self.L.ht_stage_multistep(self.h, _f(analysis), B))

    def stage_multistep_range(self, analysis, B, from_step, to_step):
        """steps [from_step, to_step) of MultiStepSim alone, from othermodel's current state"""
        analysis = _c(analysis, np.float32).reshape(B, ANALYSIS)
        self._chk(self.L.ht_stage_multistep_range(self.h, _f(analysis), B, int(from_step), int(to_step)))

    def stage_scratch_unibody(self, analysis, B, n_unibody):
        analysis = _c(analysis, np.float32).reshape(B, ANALYSIS)
        self._chk(self.L.ht_stage_scratch_unibody(self.h, _f(analysis), B, n_unibody))

    # -- profiling
    def profile_enable(self, level=1):
        """0 off; 1 bracket only the dominant kernel (k_solve) with HIP events; 2 bracket every phase (serialises the side streams)."""
        self._chk(self.L.ht_profile_enable(self.h, int(level)))

    def profile_read(self, reset=True):
        names = C.create_string_buffer(64 * 48); ms = np.zeros(64, np.float32); n = np.zeros(64, np.int32); k = C.c_int()
        self._chk(self.L.ht_profile_read(self.h, int(reset), 64, names, 48, _f(ms), _i(n), C.byref(k)))
        out = {}
        for j in range(k.value):
            out[names.raw[48 * j:48 * (j + 1)].split(b"\0", 1)[0].decode()] = (float(ms[j]), int(n[j]))
        return out

    def debug_solve_stats(self, B, reset=True):
        """Per-frame k_solve statistics [B,16] (only filled when HT_DEBUG_SKIP=2048 is set in the environment)."""
        out = np.zeros((B, 16), np.float32)
        self._chk(self.L.ht_debug_solve_stats(self.h, int(B), _f(out), int(reset)))
        return out

    # ---- multi-GPU result gather (RCCL all-gather on the context's communication stream) ----
    def comm_init(self, world, rank, unique_id):
        """Join the RCCL communicator made from `unique_id` (128 bytes from comm_unique_id() on rank 0)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.L.ht_comm_init(self.h, int(world), int(rank), C.cast(buf, C.c_void_p)))

    def comm_info(self):
        w, r = C.c_int(0), C.c_int(0)
        self._chk(self.L.ht_comm_info(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def gather_poses_dev(self, d_local, d_all, frames, slot, stream=None):
        self._chk(self.L.ht_gather_poses_dev(self.h, C.c_void_p(d_local), C.c_void_p(d_all), int(frames), int(slot), C.c_void_p(stream or 0)))

    def gather_wait(self, slot, stream=None):
        self._chk(self.L.ht_gather_wait(self.h, int(slot), C.c_void_p(stream or 0)))

    def debug_reset_organisation(self):
        """0 / 1: the latest update launched the full-reset branch in its few-frames / many-frames organisation; -1 before the first update"""
        m = C.c_int(-1)
        self._chk(self.L.ht_debug_reset_organisation(self.h, C.byref(m)))
        return m.value

    def gather_wait_host(self, slot):
        self._chk(self.L.ht_gather_wait_host(self.h, int(slot)))

    def debug_reset_flags(self, n):
        """which tracker slots took the full-reset branch (handtrack.h:706-711) in the latest update"""
        f = np.zeros(n, np.int32)
        self._chk(self.L.ht_debug_reset_flags(self.h, f.ctypes.data_as(C.POINTER(C.c_int)), n))
        return f

    def debug_solver_build(self, which):
        """Test aid: pin k_solve's build (0 auto, 1 small, 2 only, 3 mid, 4 tiny = every row array in HBM: placement only, results identical; 5 = the exact-order
        instantiation: the reference's own sweeps, tests/test_gpu_exact_solver.py)."""
        self._chk(self.L.ht_debug_solver_build(self.h, int(which)))

    def debug_contact_kernel(self, which):
        """Test aid: pin the contact kernel's organisation (0 auto, 1 cooperative, 2 lane-per-pair).  Same contacts either way."""
        self._chk(self.L.ht_debug_contact_kernel(self.h, int(which)))

    def contact_capacity(self):
        """(samples_bound, patches_bound, pool, patch_slots): the most touching samples / five-sample patches a frame of this model can produce, and what the contact kernel holds"""
        v = [C.c_int() for _ in range(4)]
        self._chk(self.L.ht_contact_capacity(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def debug_solve_tables(self, on):
        """Test aid: 0 = k_solve makes every solve's tables in its own prologue (as until round 5), 1 = k_solve_prep makes them beside the contact kernel.  Same results."""
        self._chk(self.L.ht_debug_solve_tables(self.h, int(on)))

    def debug_solve_tables_header(self, B):
        """Tuning aid: the header words [B,32] of the frames' solve tables as the latest k_solve_prep left them."""
        h = np.zeros((B, 32), np.int32)
        self._chk(self.L.ht_debug_solve_tables_header(self.h, B, _i(h)))
        return h

    def debug_contact_stats(self, B, reset=True):
        """Per-frame k_contacts statistics [B,12] + the polytope runs of the frame's wave [B,5] (only filled when HT_DEBUG_SKIP=2048 is set in the environment)."""
        out = np.zeros((B, 24), np.float32)
        self._chk(self.L.ht_debug_contact_stats(self.h, int(B), _f(out), int(reset)))
        return np.concatenate([out[:, 12:], out[:, :5]], axis=1)

    def segment_vr(self, depth, cams, entry_options=0xF, wrange=(0.1, 0.65), diam=0.17):
        """HandSegmentVR (handtrack.h:280-344) for a batch: depth u16[B,h,w], cams [B,12] -> (tiles u16[B,64,64], cams [B,12])."""
        depth = _c(depth, np.uint16); B, h, w = depth.shape
        cams = _c(cams, np.float32).reshape(B, CAM)
        tiles = np.empty((B, 64, 64), np.uint16); co = np.empty((B, CAM), np.float32)
        self._chk(self.L.ht_segment_vr(self.h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), _f(cams), w, h, B, int(entry_options), float(wrange[0]), float(wrange[1]), float(diam),
                                       tiles.ctypes.data_as(C.POINTER(C.c_uint16)), _f(co)))
        return tiles, co

    def scale(self, s):
        """HandTracker::scale (handtrack.h:591): both models of every slot grow by the factor s."""
        self._chk(self.L.ht_scale(self.h, float(s)))

    def slowfit(self, B, hold=0, refpose=None, steps=6, select_rb=-1, spoint=None, rbpoint=None, crays=None):
        """HandTracker::slowfit (handtrack.h:786-821) on the handmodel of slots [0,B) with the points of the last stage_prepare."""
        ref = None if refpose is None else _c(refpose, np.float32)
        sp = None if spoint is None else _c(spoint, np.float32); rp = None if rbpoint is None else _c(rbpoint, np.float32)
        cr = None if crays is None else _c(crays, np.float32)
        self._chk(self.L.ht_slowfit(self.h, int(B), int(hold), None if ref is None else _f(ref), int(steps), int(select_rb), None if sp is None else _f(sp), None if rp is None else _f(rp),
                                    None if cr is None else _f(cr), 0 if cr is None else int(cr.reshape(-1, 8, 4).shape[1])))

    def cnn_train(self, inputs, targets, alpha=0.001):
        """CNN::Train (cnn.h:558-580) sample after sample; returns the per-sample mean squared errors."""
        x = _c(inputs, np.float32).reshape(-1, CNN_IN); t = _c(targets, np.float32).reshape(-1, CNN_OUT)
        mse = np.zeros(x.shape[0], np.float32)
        self._chk(self.L.ht_cnn_train(self.h, _f(x), _f(t), x.shape[0], float(alpha), _f(mse)))
        return mse

    def cnn_get_weights(self):
        w = np.empty(9458400, np.float32)
        self._chk(self.L.ht_cnn_get_weights(self.h, _f(w), w.size))
        return w


def comm_available():
    """True when RCCL can be loaded on this host (ht_comm_available): agreed on by all ranks before any of them calls comm_init."""
    return load().ht_comm_available() == 1


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 makes it, the host program distributes it)."""
    L = load()
    buf = C.create_string_buffer(128)
    if L.ht_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
        raise HTError("ht_comm_unique_id failed (RCCL not available?)")
    return buf.raw


def expected_cnn(pose, cam):
    """GatherHandExpectedCNN(pose, camsub(cam, 4)).cnn_expected (handtrack.h:160-173), host only."""
    out = np.empty(CNN_OUT, np.float32)
    r = load().ht_expected_cnn(_f(_c(pose, np.float32)), _f(_c(cam, np.float32)), _f(out))
    if r != 0:
        raise HTError("ht_expected_cnn failed with status %d" % r)
    return out
