#!/bin/bash
mkdir -p gpurun_out/sess
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30
bash tools/power_trace.sh gpurun_out/sess/power_trace.txt
