#!/bin/bash
for v in 0 3 0 3; do
AC_GEMM_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('variant $v', round(j['value']), j['stages_ms'])"
done
