#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python tools/soak.py ) > $O/soak.txt 2>&1; tail -25 $O/soak.txt
