#!/bin/bash
# Build the plain-C example against the in-tree library (gcc, no hipcc needed for the caller).
set -e
cd "$(dirname "$0")"
LIB=$(cd ../i2sdf_amd/lib && pwd)
gcc -O2 -std=c99 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../include sdf_volume.c -o sdf_volume \
    -L/opt/rocm/lib -lamdhip64 -L"$LIB" -li2sdf_hip -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$LIB" -Wl,-rpath,'$ORIGIN/../i2sdf_amd/lib'
echo "built $(pwd)/sdf_volume"
