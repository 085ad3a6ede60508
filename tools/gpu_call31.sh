#!/bin/bash
mkdir -p gpurun_out
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe 2>/dev/null && timeout 60 /tmp/lds_probe ) > gpurun_out/c31_lds_probe.log 2>&1
cat gpurun_out/c31_lds_probe.log
