#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
for b in 95 190 285 95 190; do
  timeout 900 python bench.py --no-secondary --no-cpu-baseline --batch $b --steps 10 > /tmp/b.json 2> /dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); r=d['roofline']; c=d['config']; print('batch $b:', d['value'], 'img/s; ms/step', d['ms_per_step'], 'dominant', r.get('avg_launch_ms'), 'frac', r['frac'], 'cnn', c['cnn_ms_per_batch'], 'dsac', c['dsac_ms_per_batch'])"
done
