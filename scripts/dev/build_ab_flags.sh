#!/bin/bash
# A/B helper: ab/<name>.so = the current library with the listed translation units recompiled with extra hipcc flags
#   scripts/dev/build_ab_flags.sh libivid_iilp "-mllvm -amdgpu-sched-strategy=iterative-ilp" conv_igemm attn
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d); mkdir -p "$root/ab"
skip=""
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c "$root/ivid_amd/csrc/$u.hip" -o "$tmp/$u.o" &
  skip="$skip|/$u.o"
done
wait
objs=$(ls "$root"/ivid_amd/csrc/build/*.o | grep -v -E "${skip:1}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab/$name.so" $objs "$tmp"/*.o
rm -rf "$tmp"; echo "$root/ab/$name.so"
