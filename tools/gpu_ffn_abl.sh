for d in 3 35 67 99 131 227; do echo "LIDK_FFN_DBG=$d"; LIDK_FFN_DBG=$d timeout -k 10 100 python tools/ffn_bench.py 2>&1 | grep "fused lidk_ffn_fwd"; done
