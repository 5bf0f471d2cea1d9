#!/bin/bash
# tools/bake_assets.sh -- (re)makes the product's model assets with the product's own model builder (ht_model_bake = PhysModel::PhysModel + LoadHandModel,
# physmodel.h:444-475, handtrack.h:347-366, host code, no device needed):
#   hand_tracking_samples_amd/assets/model_hand17.htfx   from the reference's assets/model_hand.json
#   hand_tracking_samples_amd/assets/model_hand26.htfx   from the 26-bone hand of BASELINE configs[4] (tests/golden/make_model_hand26.py)
# Build container only (needs /root/reference/assets/model_hand.json).  bench.py and __graft_entry__.smoke() run on these; tests/test_model_build.py checks
# them array by array against what the reference's own constructor built (tests/golden/model_hand*.htfx).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
[ -f "$REF/assets/model_hand.json" ] || { echo "reference tree not present at $REF"; exit 2; }
cd "$ROOT"
mkdir -p hand_tracking_samples_amd/assets
T=$(mktemp -d /tmp/bake_assets.XXXXXX)
python3 tests/golden/make_model_hand26.py $T/model_hand26.json > /dev/null
python3 - "$REF/assets/model_hand.json" "$T/model_hand26.json" <<'PY'
import sys
sys.path.insert(0, ".")
from hand_tracking_samples_amd import native
native.model_bake(sys.argv[1], "hand_tracking_samples_amd/assets/model_hand17.htfx", True)
native.model_bake(sys.argv[2], "hand_tracking_samples_amd/assets/model_hand26.htfx", True)
PY
rm -rf "$T"
ls -la hand_tracking_samples_amd/assets
