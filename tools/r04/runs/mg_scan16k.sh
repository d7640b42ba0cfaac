#!/bin/bash
for f in tools/micro/lib_ablate/libeuler_hip_mg_*.so; do EULER_HIP_LIB=$PWD/$f timeout 400 python tools/r04/mg_scan16k.py 2>&1 | tail -1; EULER_HIP_LIB=$PWD/$f timeout 200 python tools/r04/mg_scan.py 2>&1 | tail -1; done
