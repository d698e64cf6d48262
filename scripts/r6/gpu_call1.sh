#!/bin/bash
# round 6, call 1: fused-kernel phase timelines of the IVID_EXP variants + igemm baseline on the 16^2 / 8^2 shapes
mkdir -p gpurun_out/r6
python scripts/r6/fused_ab.py ab/libivid_x0_tl.so ab/libivid_x1_tl.so ab/libivid_x3_tl.so ab/libivid_x0_tl.so > gpurun_out/r6/call1_fused_ab.jsonl 2>&1
DTYPE=bf16 CFGS=1,2,6 SHAPES_ONLY=8,9,10,11,12 python scripts/conv_bench.py > gpurun_out/r6/call1_igemm.jsonl 2>&1
tail -40 gpurun_out/r6/call1_fused_ab.jsonl
