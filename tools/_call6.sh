mkdir -p gpurun_out/r3l; export TMPDIR=/tmp
O=gpurun_out/r3l
for r in 1 2; do
for v in base pend2; do
  timeout 200 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs Pendulum-v1 --n 524288 --steps 4096 --chunk 256 --modes fused,fused-final 2>/dev/null | grep '^{' >> $O/e_ab.jsonl
done
for v in base mce1; do
  timeout 200 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs MountainCar-v0,MountainCarContinuous-v0 --n 524288 --steps 4096 --chunk 256 --modes fused,fused-final 2>/dev/null | grep '^{' >> $O/e_ab.jsonl
done
done
python3 - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r3l/e_ab.jsonl'):
    j=json.loads(l); d[(j['env'],j['mode'],j['tag'])].append(j['us_per_step'])
for k in sorted(d): print(k, d[k])
PY
