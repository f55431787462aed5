#!/bin/bash
# usage: tools/ab_lib.sh tools/_ab/libA.so tools/_ab/libB.so ...  -- quick bench.py runs with the in-tree library (first and last) and each variant
P=multimodal-brain-pattern-identification_xai_amd/libbrainxai.so
cp $P /tmp/lib_base.so
run() { timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gradcam 2>/dev/null > /tmp/ab.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/ab.json').readline()); print('$1', d['value'], d['ms_per_step'])"; }
run base
for v in "$@"; do cp $v $P; run $v; run $v; done
cp /tmp/lib_base.so $P; run base
