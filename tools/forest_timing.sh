#!/bin/bash
# where kd_median_build's time goes (diagnostic build of the library with SMHIP_KD_TIMING: workgroup 0 prints its phases), then the normal build back
# usage: tools/forest_timing.sh ["extra hipcc flags" ...]   (one diagnostic build and run per argument; none = the defaults)
[ $# -eq 0 ] && set -- ""
for f in "$@"; do
  echo "flags: $f"
  SMHIP_EXTRA_HIPCC_FLAGS="-DSMHIP_KD_TIMING $f" python -m staticmapping_amd.build --force > /dev/null 2>&1
  python tools/forest_probe.py scans=256 2>&1 | tail -2
done
python -m staticmapping_amd.build --force > /dev/null 2>&1
