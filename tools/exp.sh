D="SPMM_LIB_PATH=build/diag/libspmm_hip.so SPMM_BENCH_NOCHECK=1"
tools/runvar.sh d4 22 $D SPMM_WIN_DBG=4 && tools/runvar.sh d4_512 22 $D SPMM_WIN_DBG=516 && tools/runvar.sh d4_1024 22 $D SPMM_WIN_DBG=1028 && tools/runvar.sh d4_2048 22 $D SPMM_WIN_DBG=2052 && tools/runvar.sh d4_3072 22 $D SPMM_WIN_DBG=3076
