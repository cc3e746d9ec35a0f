#!/bin/bash
# scratch/build_variant.sh NAME SRC.hip FLAGS... : recompile one translation unit with extra flags and link a variant library
# scratch/variants/libamt_hip_NAME.so from it and the production objects (select with AMT_HIP_LIB).
set -e
name=$1; src=$2; shift 2
cd /root/repo
mkdir -p scratch/variants
obj=scratch/variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-result "$@" -c audio-to-midi_amd/csrc/$src -o $obj
others=$(ls audio-to-midi_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/libamt_hip_$name.so $obj $others
rm -f $obj
ls -la scratch/variants/libamt_hip_$name.so
