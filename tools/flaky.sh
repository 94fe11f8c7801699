for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "error or tiny or multiple or shard or fused" 2>&1 | tail -25 | grep -vE "^\.+|^$" | head -40
  echo "--- run $i done"
done
