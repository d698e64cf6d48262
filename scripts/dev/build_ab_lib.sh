#!/bin/bash
# A/B helper: build ab/<name>.so = the current library with ONE translation unit taken from another git revision
# (or from a file path), so that a kernel change can be measured on the same box against its predecessor:
#   scripts/dev/build_ab_lib.sh HEAD conv_igemm libivid_base      -> ab/libivid_base.so
# then e.g.  LIBS="- ab/libivid_base.so" bash scripts/gpu_ab.sh   (bench.py honours IVID_HIP_LIB)
set -e
rev=$1; unit=$2; name=$3
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/ivid_amd/csrc" "$tmp/include" "$root/ab"
for f in common.h internal.h ${unit}.hip; do git -C "$root" show "$rev:ivid_amd/csrc/$f" > "$tmp/ivid_amd/csrc/$f"; done
git -C "$root" show "$rev:include/ivid_hip.h" > "$tmp/include/ivid_hip.h"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c "$tmp/ivid_amd/csrc/${unit}.hip" -o "$tmp/${unit}.o"
objs=$(ls "$root"/ivid_amd/csrc/build/*.o | grep -v "/${unit}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab/${name}.so" $objs "$tmp/${unit}.o"
rm -rf "$tmp"
echo "$root/ab/${name}.so"
