#!/bin/bash
# tile sweep on the 1x1 launches (are the auto picks still the best after the skewed issue?)
mkdir -p gpurun_out/r6
CFGS=0,1,2,6 SHAPES_ONLY=7,13,14,15,16,17,2 bash scripts/r6/gpu_igemm_ab.sh ab/libivid_x0.so > gpurun_out/r6/call15_tiles_1x1.log 2>&1
