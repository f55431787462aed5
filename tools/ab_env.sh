#!/bin/bash
# usage: tools/ab_env.sh "VAR1=a VAR2=b" "VAR1=c" ...   -- one quick bench.py run per environment setting (same box, back to back)
for setting in "$@"; do
  env $setting timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gradcam 2>/dev/null > /tmp/ab.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/ab.json').readline()); print('$setting', d['value'], d['ms_per_step'])"
done
