timeout 200 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k m256 2>&1 | tail -2
for a in 0 3 2; do MSGL_M256_ABLATE=$a timeout 60 python tools/_steps.py 2>&1 | grep ABL; done
