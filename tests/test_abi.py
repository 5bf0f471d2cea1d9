"""CPU-only checks of the drop-in boundary: the shared library builds, loads and exports every symbol the header declares,
entry points fail loudly without a GPU, and the product does not depend on the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ht_mi355x.h")).read()
    return sorted(set(re.findall(r"\b(?:int|const char \*)\s*\*?\s*(ht_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported():
    from hand_tracking_samples_amd import native
    L = native.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert sorted(native.SYMBOLS) == declared, "native.SYMBOLS out of sync with include/ht_mi355x.h"
    nm = subprocess.run(["nm", "-D", "--defined-only", native.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in nm.splitlines() if l.strip())
    for s in declared:
        assert s in exported, s
        assert hasattr(L, s)


def test_product_does_not_link_oracle():
    from hand_tracking_samples_amd import native
    nm = subprocess.run(["nm", "-D", native.lib_path()], capture_output=True, text=True, check=True).stdout
    assert " ho_" not in nm
    ldd = subprocess.run(["ldd", native.lib_path()], capture_output=True, text=True).stdout
    assert "ht_oracle" not in ldd
    for root, _, files in os.walk(os.path.join(ROOT, "hand_tracking_samples_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "ht_oracle" not in src and "oracle_lib" not in src and "libht_oracle" not in src, os.path.join(root, f)


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from hand_tracking_samples_amd import native
    with pytest.raises(native.HTError, match="no HIP device|no CPU fallback|gfx950"):
        native.Context(os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), 4)


def test_null_and_bad_arguments_are_rejected():
    from hand_tracking_samples_amd import native
    L = native.load()
    assert L.ht_destroy(None) != 0
    assert L.ht_create(None, 1, 0, None) != 0
    h = C.c_void_p()
    assert L.ht_create(b"/nonexistent/model.htfx", 0, 0, C.byref(h)) != 0


def test_params_struct_matches_header_order():
    """ctypes mirror of ht_params: same field names, in the header's order."""
    from hand_tracking_samples_amd import native
    text = open(os.path.join(ROOT, "include", "ht_mi355x.h")).read()
    body = text[text.index("typedef struct ht_params"):text.index("} ht_params;")]
    names = []
    for line in body.splitlines():
        line = line.split("/*")[0].strip()
        m = re.match(r"(?:float|int)\s+(.*);", line)
        if m:
            names += [n.strip() for n in m.group(1).split(",")]
    assert names == [f[0] for f in native.Params._fields_]


def test_weights_generator_layout():
    from hand_tracking_samples_amd import weights as W
    w = W.make_cnnb()
    assert w.dtype == np.float32 and w.size == W.CNNB_COUNT == 400 + 16 + 16384 + 64 + 2304 * 2048 + 2048 + 2048 * 2304 + 2304
    assert abs(float(np.abs(w[:400]).max()) - np.sqrt(6.0 / 425.0)) < 0.01
    assert np.array_equal(w, W.make_cnnb())
    assert not np.array_equal(w, W.make_cnnb(seed=1))


def test_cxx_compat_header_compiles_and_links(tmp_path):
    """include/ht_handtrack.hpp (HandTracker::update / CNN::Eval surface) builds with plain g++ against the C-ABI library."""
    from hand_tracking_samples_amd import native
    native.load()
    exe = str(tmp_path / "compat")
    libdir = os.path.dirname(native.lib_path())
    subprocess.check_call(["g++", "-std=c++14", "-Wall", os.path.join(ROOT, "tests", "cxx_compat_example.cpp"), "-o", exe, "-L" + libdir, "-lht_mi355x", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "model_hand17.htfx")], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stdout      # fails loudly, no fallback


@pytest.mark.gpu
def test_cxx_compat_example_runs_on_the_device(tmp_path):
    """The reference-named C++ surface end to end: CNN::Eval, HandTracker::update on a 64x64 and on a 128x128 frame, CNN::Train, saveb."""
    from hand_tracking_samples_amd import native, weights as W
    native.load()
    exe = str(tmp_path / "compat")
    libdir = os.path.dirname(native.lib_path())
    subprocess.check_call(["g++", "-std=c++14", "-Wall", os.path.join(ROOT, "tests", "cxx_compat_example.cpp"), "-o", exe, "-L" + libdir, "-lht_mi355x", "-Wl,-rpath," + libdir])
    cnnb = str(tmp_path / "w.cnnb")
    W.save_cnnb(cnnb, W.make_cnnb())
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "model_hand17.htfx"), cnnb], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0
    assert "bones=17" in r.stdout and "full frame bones=17 cnn_input 64x64" in r.stdout and "train mse=" in r.stdout and "saved=37833600" in r.stdout
    assert "rows: fit + update ok" in r.stdout      # PhysModel::FitPointCloud(points, linears, angulars, microforce) and PhysicsUpdate() with caller-built rows


def test_config_read_follows_the_reference_decoder(tmp_path):
    """load_config (handtrack.h:822-828): listed fields are assigned from the file, a missing member reads as 0, a missing file is ignored."""
    from hand_tracking_samples_amd import native
    L = native.load()
    L.ht_config_read.argtypes = [C.c_char_p, C.POINTER(native.Params), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    p = native.Params(); p.microforce = 9.0; p.steps = 5; p.drangey = 0.7
    seg, pfe = C.c_float(0.17), C.c_float(1.5)
    assert L.ht_config_read(str(tmp_path / "absent.json").encode(), C.byref(p), C.byref(seg), C.byref(pfe)) == 0
    assert p.microforce == 9.0 and p.steps == 5 and seg.value == C.c_float(0.17).value and pfe.value == 1.5
    cfg = tmp_path / "config.json"
    cfg.write_text('{"microforce": 3, "drangey": 0.65, "steps": 4.9, "always_take_cnn": 1, "segment_scale": 0.2, "comment": "x", "physics_iterations": 12}')
    assert L.ht_config_read(str(cfg).encode(), C.byref(p), C.byref(seg), C.byref(pfe)) == 0
    assert p.microforce == 3.0 and abs(p.drangey - 0.65) < 1e-7 and p.steps == 4 and p.always_take_cnn == 1 and p.physics_iterations == 12
    assert abs(seg.value - 0.2) < 1e-7
    assert p.mainthreadpasses == 0 and p.min_point_num == 0 and pfe.value == 0.0      # not in the file: the reference's decoder reads 0
    vox = tmp_path / "voxel.json"; vox.write_text('{"microforce": 3, "subsample_voxel": 1, "subsample_size": 0.01}')
    assert L.ht_config_read(str(vox).encode(), C.byref(p), C.byref(seg), C.byref(pfe)) == 0
    assert p.subsample_voxel == 1 and abs(p.subsample_size - 0.01) < 1e-9      # the voxel option (physmodel.h:66-118) is read like every other field
    bad = tmp_path / "bad.json"; bad.write_text('{"microforce": ')
    assert L.ht_config_read(str(bad).encode(), C.byref(p), C.byref(seg), C.byref(pfe)) != 0


def test_dataset_formats_round_trip(tmp_path):
    """include/ht_formats.hpp: .rs/.ir/.pose/.json written like DepthDataStreamOut (dataset.h:62-104) and read back like load_dataset / LoadAnimBank."""
    lib = os.path.join(ROOT, "hand_tracking_samples_amd")
    exe = str(tmp_path / "formats")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx_formats_example.cpp"), "-o", exe,
                           "-L" + lib, "-lht_mi355x", "-Wl,-rpath," + lib])
    args = [exe, str(tmp_path / "set0")]
    ref_bank = "/root/reference/assets/animbank.pose"
    if os.path.exists(ref_bank):
        args.append(ref_bank)      # the reference's own animation bank parses to 2336 rows of 17 poses
    out = subprocess.check_output(args).decode()
    assert out.strip().endswith("OK"), out
