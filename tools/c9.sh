mkdir -p gpurun_out
timeout 500 python tools/small_batch_ab.py --out gpurun_out/c9_small_batch_ab.json 2>&1 | grep -v "^\[gemm_tune\]\|amdgpu.ids" | cut -c1-260 | tail -40
