"""ctypes binding of the CPU oracle (oracle/libht_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL = os.path.join(GOLDEN, "model_hand17.htfx")
_LIB = os.path.join(ROOT, "oracle", "libht_oracle.so")

MAXB, MAXJ = 32, 32


class F2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class F3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class F4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Pose(C.Structure):
    _fields_ = [("position", F3), ("orientation", F4)]


class Camera(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("focal", F2), ("principal", F2), ("depth_scale", C.c_float), ("pose", Pose)]


class Analysis(C.Structure):
    _fields_ = [("crays", F4 * 8), ("image_points", F2 * 8), ("confidence", C.c_float * 8), ("vals", C.c_float * 16),
                ("wristroll", C.c_float), ("pitch", C.c_float), ("tilt", C.c_float), ("palmq", F4), ("finger_clenched", C.c_float * 5)]


class Linear(C.Structure):
    _fields_ = [("rb0", C.c_int), ("rb1", C.c_int), ("position0", F3), ("position1", F3), ("normal", F3), ("targetdist", C.c_float),
                ("targetspeednobias", C.c_float), ("forcelimit", F2), ("friction_master", C.c_int), ("targetspeed", C.c_float), ("impulsesum", C.c_float)]


class Angular(C.Structure):
    _fields_ = [("rb0", C.c_int), ("rb1", C.c_int), ("axis", F3), ("torque", C.c_float), ("targetspin", C.c_float), ("mintorque", C.c_float), ("maxtorque", C.c_float)]


class Contact(C.Structure):
    _fields_ = [("rb0", C.c_int), ("rb1", C.c_int), ("normal", F3), ("p0w", F3), ("p1w", F3), ("separation", C.c_float), ("p0", F3), ("p1", F3)]


class GjkContact(C.Structure):
    _fields_ = [("normal", F3), ("p0w", F3), ("p1w", F3), ("impact", F3), ("separation", C.c_float), ("dist", C.c_float), ("type", C.c_int)]


class Params(C.Structure):
    _fields_ = [("segment_scale", C.c_float), ("full_reset_on_error", C.c_float), ("angles_only", C.c_int), ("always_take_cnn", C.c_int), ("drangey", C.c_float),
                ("boundary_planes", C.c_int), ("microforce", C.c_float), ("cloudforce_max_point", C.c_float), ("cloudforce_max_sum", C.c_float),
                ("mainthreadpasses", C.c_int), ("subsample_fraction", C.c_int), ("min_point_num", C.c_int), ("accum_error_threshold", C.c_float), ("min_cray_prob", C.c_float),
                ("steps", C.c_int), ("steps_keypoints", C.c_int), ("steps_keyangles", C.c_int), ("steps_palmangle", C.c_int), ("steps_cloudstart", C.c_int), ("steps_unibody", C.c_int),
                ("subsample_voxel", C.c_int), ("subsample_size", C.c_float)]


class Physics(C.Structure):
    _fields_ = [("deltaT", C.c_float), ("restitution", C.c_float), ("gravity", F3), ("coloumb", C.c_float), ("biasfactorjoint", C.c_float), ("biasfactorpositive", C.c_float),
                ("biasfactornegative", C.c_float), ("falltime_to_ballistic", C.c_float), ("driftmax", C.c_float), ("damping", C.c_float),
                ("iterations", C.c_int), ("iterations_post", C.c_int), ("use_collision", C.c_int), ("weak_force", C.c_float), ("bone_sum_error_scale", C.c_float), ("unibody_force", C.c_float)]


class TrackerHead(C.Structure):
    """Leading members of ho_tracker (phys, par); the rest is opaque."""
    _fields_ = [("phys", Physics), ("par", Params)]


def build():
    """(Re)build the oracle with its Makefile; cheap when up to date."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        L.ho_create.restype = C.c_void_p; L.ho_create.argtypes = [C.c_char_p]
        L.ho_destroy.argtypes = [C.c_void_p]
        L.ho_load_weights.argtypes = [C.c_void_p, fp, C.c_size_t]; L.ho_load_weights.restype = C.c_int
        L.ho_set_round_once.argtypes = [C.c_int]; L.ho_set_round_once.restype = None
        L.ho_set_cnn_override.argtypes = [C.c_void_p, fp]; L.ho_set_cnn_override.restype = None
        L.ho_set_trace.argtypes = [C.c_void_p, fp]; L.ho_set_trace.restype = None
        L.ho_get_analysis.argtypes = [C.c_void_p, fp]; L.ho_get_analysis.restype = None
        L.ho_set_direct.argtypes = [C.c_void_p, C.c_int, fp, C.c_size_t]; L.ho_set_direct.restype = C.c_int
        L.ho_set_state.argtypes = [C.c_void_p, C.c_int, fp]; L.ho_get_state.argtypes = [C.c_void_p, C.c_int, fp]
        L.ho_set_pose.argtypes = [C.c_void_p, C.c_int, fp]; L.ho_reset_tracker.argtypes = [C.c_void_p, fp]
        L.ho_get_flags.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]; L.ho_get_flags.restype = None
        L.ho_cnn_eval.argtypes = [fp, fp, fp, C.POINTER(fp)]
        L.ho_cnn_eval_sized.argtypes = [fp, C.c_int, fp, fp, C.POINTER(fp)]
        L.ho_cnn_input.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_float, C.c_float, C.c_float, fp]
        L.ho_decode.argtypes = [fp, C.POINTER(Camera), C.POINTER(Analysis)]
        L.ho_pointcloud.argtypes = [C.POINTER(C.c_uint16), C.POINTER(Camera), C.c_float, C.c_float, C.c_int, C.POINTER(F3), C.c_int, C.POINTER(C.c_int)]; L.ho_pointcloud.restype = C.c_int
        L.ho_fit_error.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(F3), C.c_int, C.POINTER(C.c_uint16), C.POINTER(Camera)]; L.ho_fit_error.restype = C.c_float
        L.ho_closest.argtypes = [C.c_void_p, F3, C.POINTER(F4)]; L.ho_closest.restype = C.c_int
        L.ho_cloud_constraint.argtypes = [C.c_void_p, F3, F3]; L.ho_cloud_constraint.restype = Linear
        L.ho_enhancements.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Angular), C.POINTER(C.c_int), C.c_int, F3, F3, C.c_int]
        L.ho_joint_linears.argtypes = [C.c_void_p, C.POINTER(Linear)]; L.ho_joint_linears.restype = C.c_int
        L.ho_joint_angulars.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Angular)]; L.ho_joint_angulars.restype = C.c_int
        L.ho_apply_angles.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Analysis), Pose, C.c_float, C.c_float, C.POINTER(Angular)]; L.ho_apply_angles.restype = C.c_int
        L.ho_cloud_chamber.argtypes = [C.c_void_p, C.POINTER(F3), C.c_int, C.POINTER(Linear), C.c_float]; L.ho_cloud_chamber.restype = C.c_int
        L.ho_find_contacts.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Contact), C.c_int]; L.ho_find_contacts.restype = C.c_int
        L.ho_fit_pointcloud.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(F3), C.c_int, C.POINTER(Linear), C.c_int, C.POINTER(Angular), C.c_int, C.c_float]
        L.ho_multistep.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Analysis), C.POINTER(F3), C.c_int, Pose]
        L.ho_pose_from_scratch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(F3), C.c_int, C.POINTER(Analysis), Pose]
        L.ho_unibody_fit.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(F3), C.c_int, F3]
        L.ho_update_cnn_model.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(Camera), fp]; L.ho_update_cnn_model.restype = C.c_int
        L.ho_update.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(Camera), fp]
        L.ho_camera_from12.argtypes = [fp, C.c_int, C.c_int, C.POINTER(Camera)]
        L.ho_model_ptr.argtypes = [C.c_void_p, C.c_int]; L.ho_model_ptr.restype = C.c_void_p
        L.ho_contact_patch_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(GjkContact)]; L.ho_contact_patch_bodies.restype = C.c_int
        L.ho_separated_bodies.argtypes = [C.c_void_p, C.c_void_p]; L.ho_separated_bodies.restype = GjkContact
        L.ho_body_ptr.argtypes = [C.c_void_p, C.c_int]; L.ho_body_ptr.restype = C.c_void_p
        L.ho_scale.argtypes = [C.c_void_p, C.c_float]
        L.ho_cnn_train.argtypes = [fp, fp, fp, C.c_float]; L.ho_cnn_train.restype = C.c_float
        L.ho_expected_cnn.argtypes = [fp, C.POINTER(Camera), fp, fp]
        L.ho_slowfit.argtypes = [C.c_void_p, C.POINTER(F3), C.c_int, C.c_int, fp, C.c_int, C.c_int, F3, F3, fp, C.c_int]
        L.ho_segment_vr.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int, fp, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint16), fp, C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)]; L.ho_segment_vr.restype = C.c_int
        _lib = L
    return _lib


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u16ptr(a):
    assert a.dtype == np.uint16 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def f3ptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.shape[-1] == 3
    return a.ctypes.data_as(C.POINTER(F3))


def v3(a):
    return F3(float(a[0]), float(a[1]), float(a[2]))


def camera(c12, w=64, h=64):
    cam = Camera()
    lib().ho_camera_from12(fptr(np.ascontiguousarray(c12, dtype=np.float32)), w, h, C.byref(cam))
    return cam


def linears_to_array(rows, n):
    """[n,16] in the layout ref_harness dumps: rb0 rb1 position0 position1 normal targetdist targetspeednobias forcelimit(2) friction_master."""
    out = np.zeros((n, 16), dtype=np.float32)
    for i in range(n):
        r = rows[i]
        out[i] = [r.rb0, r.rb1, r.position0.x, r.position0.y, r.position0.z, r.position1.x, r.position1.y, r.position1.z, r.normal.x, r.normal.y, r.normal.z,
                  r.targetdist, r.targetspeednobias, r.forcelimit.x, r.forcelimit.y, r.friction_master]
    return out


def angulars_to_array(rows, n):
    out = np.zeros((n, 8), dtype=np.float32)
    for i in range(n):
        r = rows[i]
        out[i] = [r.rb0, r.rb1, r.axis.x, r.axis.y, r.axis.z, r.targetspin, r.mintorque, r.maxtorque]
    return out


class Oracle:
    """Thin object wrapper over ho_tracker."""

    def __init__(self, weights=None, model=None):
        self.L = lib()
        model = model or MODEL
        import htfx
        self.nb = len(htfx.load(model)["nverts"])
        self.h = self.L.ho_create(model.encode())
        assert self.h, "ho_create failed"
        self.head = TrackerHead.from_address(self.h)
        if weights is not None:
            assert self.L.ho_load_weights(self.h, fptr(weights), weights.size) == 0

    def close(self):
        if self.h:
            self.L.ho_destroy(self.h)
            self.h = None

    def model(self, which):
        return self.L.ho_model_ptr(self.h, which)

    def set_state(self, which, s):
        self.L.ho_set_state(self.h, which, fptr(np.ascontiguousarray(s, dtype=np.float32)))

    def get_state(self, which):
        s = np.zeros((self.nb, 13), dtype=np.float32)
        self.L.ho_get_state(self.h, which, fptr(s))
        return s

    def flags(self):
        """(prev_frame_error, initializing, points of the last update)"""
        e, i, n = C.c_float(), C.c_int(), C.c_int()
        self.L.ho_get_flags(self.h, C.byref(e), C.byref(i), C.byref(n))
        return e.value, i.value, n.value

    def reset(self, pose7):
        self.L.ho_reset_tracker(self.h, fptr(np.ascontiguousarray(pose7, dtype=np.float32)))


def segment_vr(depth, cam12, entry_options=0xF, wrange=(0.1, 0.65), diam=0.17):
    """HandSegmentVR restatement: depth u16[h,w] + camera[12] -> (tile u16[64,64], camera[12], small u16[h/4,w/4], dt u8[h/4,w/4])."""
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    cam12 = np.ascontiguousarray(cam12, np.float32)
    tile = np.zeros((64, 64), np.uint16); cam = np.zeros(12, np.float32)
    small = np.zeros((h // 4, w // 4), np.uint16); dt = np.zeros((h // 4, w // 4), np.uint8)
    lib().ho_segment_vr(u16ptr(depth), w, h, fptr(cam12), int(entry_options), float(wrange[0]), float(wrange[1]), float(diam),
                        u16ptr(tile), fptr(cam), u16ptr(small), dt.ctypes.data_as(C.POINTER(C.c_uint8)))
    return tile, cam, small, dt


def cnn128_eval(weights128, x, layers=None):
    """The 128x128-input net (SURVEY 8d config 5 ii) on inputs x [n, 16384] -> [n, 2304]; `layers`: optional dict index -> array to fill for x[0]."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 128 * 128)
    w = np.ascontiguousarray(weights128, np.float32)
    out = np.zeros((len(x), 2304), np.float32)
    for i in range(len(x)):
        keep = None
        if layers is not None and i == 0:
            arr = (C.POINTER(C.c_float) * 11)()
            for k, a in layers.items():
                arr[k] = fptr(a)
            keep = arr
        lib().ho_cnn_eval_sized(fptr(w), 128, fptr(x[i]), fptr(out[i]), keep)
    return out
