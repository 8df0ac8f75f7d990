#!/bin/bash
# A/B builds of one unit (UNIT=n3 by default, e.g. UNIT=n3_enum): tools/ab_build.sh NAME [extra hipcc flags]  ->  build_ab/libNAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab
# (links against the other units' objects of the last regular build; never rebuilds the main library)
unit=${UNIT:-n3}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function "$@" -c theta_amd/csrc/$unit.hip -o build_ab/${unit}_$name.o
objs=""
for u in n2 n3 n3_enum n3_sieve n3_sieve_witness bnb batch api comm; do
    if [ "$u" = "$unit" ]; then objs="$objs build_ab/${unit}_$name.o"; else objs="$objs theta_amd/csrc/$u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/lib$name.so $objs
echo build_ab/lib$name.so
