#!/bin/bash
# round-4 session O: wave trace of the layer and objective kernels (tools/make_trace_build.sh first; tools/wavetrace.cpp)
set -u
mkdir -p gpurun_out
timeout 300 ./tools/wavetrace inverserenderingofindoorscene_amd/variants/libsgrender_trace.so 16 > gpurun_out/trace_o.txt 2> gpurun_out/trace_o.err
tail -3 gpurun_out/trace_o.err
python tools/wavetrace_report.py gpurun_out/trace_o.txt | tee gpurun_out/r04o_wavetrace_report.txt
rm -f gpurun_out/trace_o.txt
