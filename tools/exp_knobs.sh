# what a phase costs a 1024-frame step: the update timed with the phase switched off (upper bounds; tools/time_update.py)
for k in "" "full_reset_on_error=100" "boundary_planes=0" "physics_use_collision=0" "mainthreadpasses=1" "steps_cloudstart=100" "steps=1"; do
  timeout -k 10 120 python tools/time_update.py $k 2>&1 | tail -1
done
