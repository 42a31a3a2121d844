#!/bin/bash
# A/B of two builds on the same box: alternates bench.py runs of libssak_hip.so and libssak_hip_alt.so
for i in 1 2 3; do
  a=$(python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  b=$(SSAK_HIP_LIB=$PWD/ssak_amd/lib/libssak_hip_alt.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "current $a   alt $b"
done
