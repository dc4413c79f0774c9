# fused FeedForward kernels: parity tests, then the microbench under the ablation switches of csrc/ffn.hip
timeout -k 10 300 python -m pytest tests/test_gpu_ffn.py -x -q 2>&1 | tail -3
for d in ${DBGS:-0 1 2 3 16}; do echo "LIDK_FFN_DBG=$d"; LIDK_FFN_DBG=$d timeout -k 10 100 python tools/ffn_bench.py 2>&1 | grep "fused"; done
