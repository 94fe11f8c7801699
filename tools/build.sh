#!/bin/bash
# builds gencore_amd/csrc/libgencore_amd.so in place: tools/build.sh [extra hipcc flags]
cd "$(dirname "$0")/../gencore_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" engine.hip bamio.cpp -o libgencore_amd.so -lz -lpthread 2>&1 | grep -v "warning: ignoring return\|note:" | grep -B2 -A10 "error" | head -40
