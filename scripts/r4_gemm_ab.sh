mkdir -p gpurun_out/r4; python -m pytest tests/test_gpu_dgemm_mfma.py -x -q -p no:cacheprovider 2>&1 | tail -2
for bm in 64 128; do echo "BM=$bm"; PDMP_DGEMM_BM=$bm python3 scripts/bench_dgemm.py; done
