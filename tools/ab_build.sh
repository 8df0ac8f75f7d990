#!/bin/bash
# A/B builds of the n3 search kernel: tools/ab_build.sh NAME [extra hipcc flags]  ->  build_ab/libNAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab
# (links against the other units' objects of the last regular build; never rebuilds the main library)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function "$@" -c theta_amd/csrc/n3.hip -o build_ab/n3_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/lib$name.so theta_amd/csrc/n2.o build_ab/n3_$name.o theta_amd/csrc/batch.o theta_amd/csrc/api.o
echo build_ab/lib$name.so
