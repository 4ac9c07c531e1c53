#!/bin/bash
# Run ON THE GPU BOX: un-profiled bench lines of the BASELINE configurations (ms/step, value), for A/B measurements in one call.
for p in cfg3 cfg3 cfg2 cfg4 cfg5; do
  timeout 300 python bench.py --no-cpu-baseline --no-drop-in --preset $p "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', round(d['ms_per_step'],5), '%.4g' % d['value'])"
done
