#!/bin/bash
# config 3 end to end, host-driven loop vs every chain as one ivid_sample call; and the opt-in ladder for comparison
mkdir -p gpurun_out/r6
IVID_DEVICE_LOOP=1 timeout 1500 python bench.py --config c3 > gpurun_out/r6/bench_c3_device_loop.json 2> gpurun_out/r6/bench_c3_device_loop.err
echo "c3 device loop exit $?"; head -c 300 gpurun_out/r6/bench_c3_device_loop.json; echo
timeout 1500 python bench.py --config c3 --precision fp16sa3 > gpurun_out/r6/bench_c3_fp16sa3.json 2> gpurun_out/r6/bench_c3_fp16sa3.err
echo "c3 fp16sa3 exit $?"; head -c 300 gpurun_out/r6/bench_c3_fp16sa3.json; echo
