#!/bin/bash
# tools/regen_goldens.sh -- rebuilds oracle/_ref/ref_harness from the reference headers where they lie (/root/reference, build container only) and
# regenerates EVERY fixture under tests/golden/ with the exact arguments it was made with, into a scratch directory, then compares each with the
# committed file (cmp for the .htfx containers, array by array for the .npz archives).  Exit status 0 = every committed fixture is reproduced.
#
#   bash tools/regen_goldens.sh            compare only
#   bash tools/regen_goldens.sh --write    also overwrite tests/golden/ with the regenerated files
#
# Fixed inputs: animation bank /root/reference/assets/animbank.pose; CNN weights from the seeded generator (hand_tracking_samples_amd/weights.py,
# restated in ref_harness): seed 0x5EED0001, FC2 gain 24.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
BANK=$REF/assets/animbank.pose
SEED=0x5EED0001
GAIN=24
[ -f "$BANK" ] || { echo "reference tree not present at $REF: fixtures can only be regenerated in the build container"; exit 2; }
make -s -C "$ROOT/oracle" ref
H=$ROOT/oracle/_ref/ref_harness
G=$ROOT/tests/golden
T=$(mktemp -d /tmp/regen_goldens.XXXXXX)
WRITE=0; [ "$1" = "--write" ] && WRITE=1
cd "$ROOT"

# models: the reference's own hand (17 bones), the 26-bone hand of BASELINE configs[4] and a 3-body chain with unusual faces, all in the reference's JSON schema
$H model $T/model_hand17.htfx
python3 tests/golden/make_model_hand26.py $T/model_hand26.json
HT_REF_MODEL_JSON=$T/model_hand26.json $H model $T/model_hand26.htfx
python3 tests/golden/make_model_chain3.py $T/model_chain3.json
$H modelfile $T/model_chain3.json $T/model_chain3.htfx
# per-stage goldens of the 64x64 path: 8 animation-bank rows (open hand, fists, P < 400 and P > 400 points, one full-reset frame), 48 GJK/EPA cases
$H golden $BANK 0,16,144,1584,2224,912,1504,2048 $SEED $GAIN $T/golden8.htfx
# bench / batch-parity input: 256 frames, rows 3, 12, 21, ... (first 3, stride 9)
$H frames $BANK 3 9 256 $T/frames256.htfx
# the reference's result of the whole unit of work on every one of those 256 frames (user poses, othermodel poses, tracker flags)
$H poses $T/frames256.htfx $SEED $GAIN $T/poses256.htfx
# the bench's batch: 1024 DISTINCT frames, rows (3 + 9 i) mod 2336 (the first 256 are the set above), and the reference's results for all of them
$H frames $BANK 3 9 1024 $T/frames1024.htfx
$H poses $T/frames1024.htfx $SEED $GAIN $T/poses1024.htfx
# the same with the application's always_take_cnn switch (synthetic-tracker.cpp:91): every frame's user pose then depends on the net, its decode and MultiStepSim
$H poses $T/frames1024.htfx $SEED $GAIN $T/poses1024_takecnn.htfx takecnn
# what the application draws (synthetic-tracker.cpp:191,204-209): DepthMesh and the heat-map label images of animation-bank row 144
$H viz $BANK 144 $T/viz1.htfx
# the optional voxel sub-sampling of the main-thread cloud (handtrack.h:535-536): 1 cm voxels, min_point_num 20
$H voxel $BANK 0,912,2224,1504 $SEED $GAIN 0.01 20 $T/voxel4.htfx
# on-disk dataset formats (dataset.h): a three-frame set written by the reference's DepthDataStreamOut, what its load_dataset returns for it, and the
# reference-held sample dataset's header and poses as the reference decodes them (the two sample files are copied as data; their .rs/.ir blobs are stripped upstream)
mkdir -p $T/dataset3
$H dataset_write $T/dataset3/ set3
$H dataset_read $T/dataset3/set3 17 $T/dataset3/set3_as_the_reference_reads_it.htfx > /dev/null
cp $REF/datasets/example/hand_data_example.json $REF/datasets/example/hand_data_example.pose $T/dataset3/
chmod u+w $T/dataset3/hand_data_example.*
$H dataset_header $T/dataset3/hand_data_example.json $T/dataset3/hand_data_example.pose 17 $T/dataset3/hand_data_example_as_the_reference_reads_it.htfx
# next rows of SURVEY 8(f)
$H segment $BANK 0,144,912,1504,2048,2224 $T/seg.htfx
$H scale $BANK 0,912 $SEED $GAIN 1.15 $T/scale_frames.htfx      # writes the tracked frames and, beside them, the scaled model (<out>.model)
python3 - "$T/scale_frames.htfx" "$T/scale115.htfx" <<'PY'
# scale115.htfx = the tracked frames followed by the scaled model's arrays under "model/"
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import htfx
from bench import _write_htfx
a = htfx.load(sys.argv[1]); m = htfx.load(sys.argv[1] + ".model")
out = dict(a); out.update({"model/" + k: v for k, v in m.items()})
_write_htfx(sys.argv[2], out)
PY
$H slowfit $BANK 0,912,2224 $T/slowfit3.htfx
$H train $BANK 0,912,2224 $SEED $GAIN 2 $T/train3.htfx
# full-size frames: the application's 320x240 camera with the 17-bone hand; BASELINE configs[4] (128x128, 26 bones) both ways the reference can run it
$H fullframe $BANK 40,1234 320,240,305 $SEED $GAIN $T/fullframe320.htfx
$H fullframe $BANK 1504,2048 320,240,900 $SEED $GAIN $T/fullframe320close.htfx      # a hand close to the lens: 7723 / 10628 points per frame
HT_REF_MODEL_JSON=$T/model_hand26.json $H fullframe $BANK 0,300,912,1500 128,128,163 $SEED $GAIN $T/fullframe5.htfx
HT_REF_MODEL_JSON=$T/model_hand26.json $H config5 $BANK 0,300,912,1500 $SEED $GAIN $T/config5.htfx
HT_REF_MODEL_JSON=$T/model_hand26.json $H fullframes $BANK 3 36 64 128,128,163 $T/frames5.htfx
HT_REF_MODEL_JSON=$T/model_hand26.json $H posesfull $T/frames5.htfx $SEED $GAIN $T/poses5full.htfx      # the reference's HandTracker on all 64 of them (bench.py --workload config5 verifies against it)
# the 128x128-input net of SURVEY 8(d) config 5 (ii), built from the reference's own layer classes, on four of those 128x128 frames
$H cnn128 $T/frames5.htfx 0,9,33,60 $SEED $GAIN $T/cnn128.htfx
# BASELINE configs[4] end to end (SURVEY 8d config 5 i-iii): that net, its decode and the tracker's stages on the 128x128 frame in one unit of work; per-stage dumps
# of four frames (a second update, a far-off start with always_take_cnn = the reset branch and the accept) and the results for all 64 frames of the bench input
HT_REF_MODEL_JSON=$T/model_hand26.json $H e2e128 $T/frames5.htfx 0,9,33,60 $SEED $GAIN $T/e2e128.htfx

# the bench's configs[4] batch: 256 distinct 128x128 frames (rows 3 + 9 i) and the reference's results for every one of them, both ways (bench.py --workload config5 / config5-e2e)
HT_REF_MODEL_JSON=$T/model_hand26.json $H fullframes $BANK 3 9 256 128,128,163 $T/frames5_256.htfx
HT_REF_MODEL_JSON=$T/model_hand26.json $H posesfull $T/frames5_256.htfx $SEED $GAIN $T/poses5full256.htfx
HT_REF_MODEL_JSON=$T/model_hand26.json $H e2e128 $T/frames5_256.htfx "" $SEED $GAIN $T/e2e128_256.htfx      # no per-stage dumps: the all/ results only

rc=0
for f in model_hand17 model_hand26 model_chain3 golden8 poses256 poses1024 poses1024_takecnn poses5full voxel4 scale115 slowfit3 train3 fullframe320 fullframe320close fullframe5 config5 cnn128 e2e128 poses5full256 e2e128_256; do
	if cmp -s $T/$f.htfx $G/$f.htfx; then echo "identical  $f.htfx"; else echo "DIFFERENT  $f.htfx"; rc=1; fi
	[ $WRITE = 1 ] && cp $T/$f.htfx $G/$f.htfx
done
for f in $(ls $T/dataset3); do
	if cmp -s $T/dataset3/$f $G/dataset3/$f; then echo "identical  dataset3/$f"; else echo "DIFFERENT  dataset3/$f"; rc=1; fi
	[ $WRITE = 1 ] && mkdir -p $G/dataset3 && cp $T/dataset3/$f $G/dataset3/$f
done
if cmp -s $T/model_chain3.json $G/model_chain3.json; then echo "identical  model_chain3.json"; else echo "DIFFERENT  model_chain3.json"; rc=1; fi
python3 - "$T" "$G" "$WRITE" <<'PY' || rc=1
import sys, numpy as np
sys.path.insert(0, "tests")
import htfx
T, G, write = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
BD = "bench_data"      # the bench's input frames (bench.py, the batch tests) live beside the tests' goldens, not among them
bad = 0
def same(name, new, old):
    global bad
    ok = set(new) == set(old) and all(np.array_equal(new[k], old[k]) for k in new)
    print(("identical  " if ok else "DIFFERENT  ") + name + " (array by array)")
    bad += not ok
f = htfx.load(T + "/frames256.htfx"); f256 = {k: f[k] for k in ("depth", "cam", "startpose", "gtpose", "rows")}
same("frames256.npz", f256, dict(np.load(G + "/frames256.npz")))
f = htfx.load(T + "/frames1024.htfx"); f1024 = {k: f[k] for k in ("depth", "cam", "startpose", "gtpose", "rows")}      # gtpose: the rendered row's own pose (labels of tools/train_synthetic.py)
same("frames1024.npz", f1024, dict(np.load(BD + "/frames1024.npz")))
g = htfx.load(T + "/frames5.htfx"); f5 = {k: g[k] for k in ("depth", "cam", "startpose", "rows")}
same("frames5_64.npz", f5, dict(np.load(G + "/frames5_64.npz")))
g = htfx.load(T + "/frames5_256.htfx"); f5b = {k: g[k] for k in ("depth", "cam", "startpose", "rows")}
same("frames5_256.npz", f5b, dict(np.load(BD + "/frames5_256.npz")))
s = htfx.load(T + "/seg.htfx"); seg = {k.replace("/", "__"): v for k, v in s.items()}
same("segment6.npz", seg, dict(np.load(G + "/segment6.npz")))
v = htfx.load(T + "/viz1.htfx"); viz = {k: (a.astype(np.uint8) if k.endswith("_labels") else a) for k, a in v.items()}      # the label images travel as 16-bit words in the container
same("viz1.npz", viz, dict(np.load(G + "/viz1.npz")))
if write:
    np.savez_compressed(G + "/frames256.npz", **f256)
    np.savez_compressed(BD + "/frames1024.npz", **f1024)
    np.savez_compressed(G + "/frames5_64.npz", **f5)
    np.savez_compressed(BD + "/frames5_256.npz", **f5b)
    np.savez_compressed(G + "/segment6.npz", **seg)
    np.savez_compressed(G + "/viz1.npz", **viz)
sys.exit(1 if bad else 0)
PY
# the tolerance yardsticks: how far the reference moves between its own IEEE and FMA-contracted builds on every bench frame (tests/golden/ref_flag_spread.py compiles the
# harness four ways for the fixed target x86-64-v3 and runs the 1024 frames with each: a few minutes), without and with always_take_cnn
python3 tests/golden/ref_flag_spread.py $T/spread.json $T/ref_spread1024.npz > /dev/null
python3 tests/golden/ref_flag_spread.py $T/spread_take.json $T/ref_spread1024_takecnn.npz takecnn > /dev/null
python3 tests/golden/ref_flag_spread.py $T/spread5.json $T/ref_spread5_256.npz config5 > /dev/null      # BASELINE configs[4]: the yardsticks bench.py --workload config5 / config5-e2e verify against
python3 tests/golden/ref_flag_spread.py $T/spread5e.json $T/ref_spread5e2e_256.npz e2e > /dev/null
python3 - "$T" "$G" "$WRITE" <<'PY' || rc=1
import sys, shutil, numpy as np
T, G, write = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
bad = 0
for f in ("ref_spread1024.npz", "ref_spread1024_takecnn.npz", "ref_spread5_256.npz", "ref_spread5e2e_256.npz"):
    a, b = dict(np.load(T + "/" + f)), dict(np.load(G + "/" + f))
    ok = set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
    print(("identical  " if ok else "DIFFERENT  ") + f + " (array by array)")
    bad += not ok
    if write: shutil.copy(T + "/" + f, G + "/" + f)
sys.exit(1 if bad else 0)
PY
rm -rf "$T"
if [ $rc = 0 ]; then echo "all fixtures reproduced"; else echo "SOME FIXTURES DIFFER"; fi
exit $rc
