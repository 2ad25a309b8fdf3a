#!/usr/bin/env bash
# A/B of libmxv build variants in ONE box (boxes differ by several %): tools/gpu_ab.sh "<variants>" "<envs>" "<modes>" [n] [rounds]
V=${1}; ENVS=${2:-CartPole-v1}; MODES=${3:-fused,fused-final}; N=${4:-1048576}; ROUNDS=${5:-2}
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for r in $(seq $ROUNDS); do for v in $V; do
  timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs $ENVS --n $N --steps 1600 --chunk 100 --modes $MODES 2>&1 | grep -v amdgpu.ids
done; done
} > gpurun_out/ab.log 2>&1
python3 - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/ab.log'):
    try: j=json.loads(l)
    except Exception: print(l.strip()); continue
    d[(j['env'],j['mode'],j['tag'])].append(j['us_per_step'])
for k in sorted(d): print(k, ' '.join(f'{x:.3f}' for x in d[k]))
PY
