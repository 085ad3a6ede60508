#!/bin/bash
mkdir -p gpurun_out
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/read_bw_probe.hip -o /tmp/read_bw && timeout 120 /tmp/read_bw ) > gpurun_out/c11_read_bw.log 2>&1
cat gpurun_out/c11_read_bw.log | cut -c1-150
