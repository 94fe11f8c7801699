for a in 1 4 16; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --align $a 2>&1 | tail -1 > /tmp/o_$a.json
  python -c "
import json
d=json.load(open('/tmp/o_$a.json')); print('align $a', d['value'], d['ms_per_step'], d['roofline']['phase_ms'])"
done
