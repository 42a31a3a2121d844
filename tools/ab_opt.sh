#!/bin/bash
# A/B on ONE box: optimizer on the side stream (default) vs in line, and the ticket tile order under the overlap
for i in 1 2; do
  SSAK_OPT_STREAM=0 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inline ', d['value'], d['ms_per_step'])"
  SSAK_OPT_STREAM=1 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side   ', d['value'], d['ms_per_step'], d['optimizer_tail']['exposed_us_per_step'])"
  SSAK_OPT_STREAM=1 SSAK_TILE_ORDER=1 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side+tk', d['value'], d['ms_per_step'], d['optimizer_tail']['exposed_us_per_step'])"
done
