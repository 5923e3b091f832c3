for k in BP0 BP5 BP6; do echo "== $k"; NEURITE_AMD_LIB=$PWD/tools/lab/libnrt_fused_$k.so WCS="1 1 1" timeout 120 python tools/bwd_wc_time.py 2>&1 < /dev/null | tail -2; done
