#!/bin/bash
mkdir -p gpurun_out/r6
python scripts/r6/persist_check.py > gpurun_out/r6/call2_persist_check.txt 2>&1; tail -15 gpurun_out/r6/call2_persist_check.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -5
FORMS=plain,LOIN,LOINres python scripts/r6/fused_ab.py ab/libivid_head.so ivid_amd/lib/libivid_hip.so ab/libivid_head.so ivid_amd/lib/libivid_hip.so > gpurun_out/r6/call2_fused_ab.jsonl 2>&1
cut -c1-200 gpurun_out/r6/call2_fused_ab.jsonl
