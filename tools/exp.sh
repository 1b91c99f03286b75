SPMM_XFER_TRACE=1 python tools/bench_preprocess.py 22 2>&1 | tail -6
