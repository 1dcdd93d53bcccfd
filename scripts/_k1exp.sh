for sc in opaque; do for d in 64 32 64 0; do echo "$sc SGR_DEBUG=$d"; SGR_DEBUG=$d timeout 120 python scripts/kernel_times.py $sc 2>&1 | grep -o '"preprocess_fwd": [0-9.]*'; done; done
