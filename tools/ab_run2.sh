#!/bin/bash
cd "$(dirname "$0")/.."
for f in build_ab/lib*fine.so; do echo "== $f"; THETA_HIP_LIB=$PWD/$f timeout 300 python tools/fine_profile.py 2>&1 | tail -3; done
build_ab/valu_rates
