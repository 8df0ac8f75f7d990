#!/bin/bash
# runs tools/phase_profile.py against every build_ab/lib*.so
cd "$(dirname "$0")/.."
for f in build_ab/lib*.so; do echo "== $f"; THETA_HIP_LIB=$PWD/$f timeout 300 python tools/phase_profile.py 2>&1 | tail -4; done
