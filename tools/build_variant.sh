#!/bin/bash
# build libgencore_amd.so of another git revision into ab/<name>.so for tools/ab.sh:  tools/build_variant.sh <git-ref> <name>
set -e
REF=$1; NAME=$2; D=$(mktemp -d)
git -C /root/repo archive "$REF" gencore_amd/csrc include | tar -x -C "$D"
mkdir -p /root/repo/abx
SRC="$D/gencore_amd/csrc/engine.hip"; [ -f "$D/gencore_amd/csrc/bamio.cpp" ] && SRC="$SRC $D/gencore_amd/csrc/bamio.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $SRC -o "/root/repo/abx/$NAME.so" -lz -lpthread
rm -rf "$D"; echo "built abx/$NAME.so from $REF"
