# fused FeedForward kernels: parity tests, then the microbench under the switches of csrc/ffn.hip (LIDK_FFN_V, LIDK_FFN_DBG)
timeout -k 10 300 python -m pytest tests/test_gpu_ffn.py -x -q 2>&1 | tail -3
for v in ${VERS:-2 1}; do for d in ${DBGS:-0 3}; do echo "LIDK_FFN_V=$v LIDK_FFN_DBG=$d"; LIDK_FFN_V=$v LIDK_FFN_DBG=$d timeout -k 10 100 python tools/ffn_bench.py 2>&1 | grep "fused"; done; done
