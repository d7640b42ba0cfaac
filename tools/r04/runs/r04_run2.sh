#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x --durations=10 > gpurun_out/r04_run2_tests.txt 2>&1
tail -15 gpurun_out/r04_run2_tests.txt
timeout 300 python tools/r04/resident_probe.py 1024 > gpurun_out/r04_resident_probe.txt 2>&1
cat gpurun_out/r04_resident_probe.txt
