timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_deep_gpu.py -q -m gpu -x 2>&1 | tail -2
for t in -1 0 4 1; do for sp in 2 4 8 16; do echo -n "tile $t splits $sp: "; RV_TILE=$t SPL_heads=$sp timeout -k 10 100 python tools/gemm_bench.py 2>&1 | grep "heads fwd"; done; done
