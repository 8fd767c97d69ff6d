#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tools/dp_overhead.py 30 2>/dev/null | tee gpurun_out/r04_dp_overhead.log
