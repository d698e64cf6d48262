#!/bin/bash
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -4
OUT=call6_epi FORMS=plain,LOIN,LOINres bash scripts/r6/gpu_ab.sh ab/libivid_x0_tl.so ab/libivid_head.so ivid_amd/lib/libivid_hip.so ab/libivid_head.so ivid_amd/lib/libivid_hip.so
