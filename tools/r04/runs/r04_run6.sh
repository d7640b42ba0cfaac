#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x 2>&1 | tail -3
timeout 300 python tools/r04/resident_probe.py 1024 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/r04/resident_probe.py 2048 2>&1 | grep -v amdgpu.ids
