#!/bin/bash
# usage: tools/ab_bench.sh VAR v1 v2 ...   -- runs bench.py once per value of the environment variable VAR
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gradcam 2>/dev/null > /tmp/ab_$v.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/ab_$v.json').readline()); print('$var=$v', d['value'], d['ms_per_step'])"
done
