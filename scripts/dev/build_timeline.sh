#!/bin/bash
# Development build of the library with phase time stamps in conv3x3_fused_kernel (-DIVID_DEV_TIMELINE) -> ab/libivid_timeline.so
# (ab/ is git-ignored but travels to the GPU box).  Never loaded by the product: scripts/dev/fused_timeline.py selects it
# through IVID_HIP_LIB.
set -e
cd "$(dirname "$0")/../.."
mkdir -p ab/obj_timeline
for f in ivid_amd/csrc/*.hip; do
  o=ab/obj_timeline/$(basename "$f" .hip).o
  if [ ! -e "$o" ] || [ "$f" -nt "$o" ] || [ ivid_amd/csrc/common.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DIVID_DEV_TIMELINE -c "$f" -o "$o" &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libivid_timeline.so ab/obj_timeline/*.o
echo ab/libivid_timeline.so
