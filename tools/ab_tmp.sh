for t in 128 192 256 320 384 512; do echo "== threads $t"; SGX_MIX_THREADS=$t python tools/quick_bench.py --others 2>&1 | grep "^W=2"; done
