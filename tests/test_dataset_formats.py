"""The on-disk dataset formats (include/dataset.h of the reference; here include/ht_formats.hpp) pinned on the reference in both directions.

tests/golden/dataset3/ holds (tools/regen_goldens.sh):
  set3.{json,rs,ir,pose,rgb,feye}                      a three-frame set written by the REFERENCE's DepthDataStreamOut (`ref_harness dataset_write`)
  set3_as_the_reference_reads_it.htfx                  everything the reference's load_dataset returns for it (`ref_harness dataset_read`)
  hand_data_example.{json,pose}                        the sample dataset the reference ships (datasets/example/; its .rs/.ir blobs are stripped there)
  hand_data_example_as_the_reference_reads_it.htfx     its header as the reference's from_json decodes it and its 69 x 17 poses (`ref_harness dataset_header`)

(1) ht_formats.hpp reads the reference's files and returns what the reference's reader returns, array by array, bit for bit.
(2) ht_formats.hpp writes the same set: the binary streams and the pose text are byte-identical to the reference's files, the header equal as JSON,
    and (where the harness binary is present) the reference's own load_dataset reads OUR files to the same result.
Host only; no device call is made."""
import json
import os
import subprocess

import numpy as np
import pytest

import htfx

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DS = os.path.join(HERE, "golden", "dataset3")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from hand_tracking_samples_amd import native
    native.load()
    lib = os.path.dirname(native.lib_path())
    out = str(tmp_path_factory.mktemp("fmt") / "formats")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "cxx_formats_example.cpp"), "-o", out,
                           "-L" + lib, "-lht_mi355x", "-Wl,-rpath," + lib])
    return out


def _same(a, b):
    assert set(a) == set(b), sorted(set(a) ^ set(b))
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


def test_reading_what_the_reference_wrote(exe, tmp_path):
    out = str(tmp_path / "read.htfx")
    subprocess.check_call([exe, "read", os.path.join(DS, "set3"), "17", out])
    ref = htfx.load(os.path.join(DS, "set3_as_the_reference_reads_it.htfx"))
    got = htfx.load(out)
    assert int(ref["nframes"][0]) == 3 and ref["f2/rgb"].size == 8 * 6 * 3 and ref["f1/fisheye"].size == 8      # the fixture really carries every stream
    _same(got, ref)


def test_the_reference_held_example_dataset(exe, tmp_path):
    out = str(tmp_path / "hdr.htfx")
    subprocess.check_call([exe, "header", os.path.join(DS, "hand_data_example.json"), os.path.join(DS, "hand_data_example.pose"), "17", out])
    ref = htfx.load(os.path.join(DS, "hand_data_example_as_the_reference_reads_it.htfx"))
    assert ref["poses"].shape == (69, 17, 7) and bytes(ref["info_camtype"].astype(np.uint8)).decode() == "ivycam"
    _same(htfx.load(out), ref)


def test_writing_what_the_reference_writes(exe, tmp_path):
    d = tmp_path / "ours"
    d.mkdir()
    subprocess.check_call([exe, "write", str(d) + "/", "set3"])
    for ext in ("rs", "ir", "rgb", "feye", "pose"):
        assert open(d / ("set3." + ext), "rb").read() == open(os.path.join(DS, "set3." + ext), "rb").read(), ext
    assert json.load(open(d / "set3.json")) == json.load(open(os.path.join(DS, "set3.json")))
    if os.path.exists(HARNESS):      # the reference's own reader on OUR files
        out = str(tmp_path / "back.htfx")
        subprocess.check_call([HARNESS, "dataset_read", str(d / "set3"), "17", out], stdout=subprocess.DEVNULL)
        _same(htfx.load(out), htfx.load(os.path.join(DS, "set3_as_the_reference_reads_it.htfx")))
    else:
        pytest.skip("oracle/_ref/ref_harness not built here: the byte comparison above stands in for the reference's reader")
