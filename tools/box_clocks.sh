#!/usr/bin/env bash
# Clocks / power UNDER LOAD (the fused CartPole rollout running) next to the achieved us/step: what differs between fast and slow boxes?
O=${1:-gpurun_out/box_clocks.txt}
python - > /tmp/load.out 2>/dev/null <<'PY' &
import sys, time, json
sys.path.insert(0, '.')
import torch
from gym_amd.rollout import DeviceRollout
import os
r = DeviceRollout(os.environ.get("BOX_ENV", "CartPole-v1"), int(os.environ.get("BOX_N", str(1 << 20))), seed=0, action_seed=1)
r.reset(seed=0)
traj = r.trajectory_buffers(256)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 6.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(r.stream)
    for _ in range(20):
        r.rollout_per_step(256, out=traj)
    e1.record(r.stream)
    r.synchronize()
    n += 1
    print(json.dumps({"t": round(time.perf_counter() - t0, 2), "us_per_step": round(e0.elapsed_time(e1) / 20 / 256 * 1e3, 3)}), flush=True)
PY
LP=$!
sleep 3.5
{
echo "== under load"
rocm-smi --showclocks --showpower --showuse --showmemuse --showtemp 2>&1 | grep -E "clock|Power|busy|Activity|Temperature|Bandwidth" 
sleep 0.7
rocm-smi --showclocks --showpower 2>&1 | grep -E "clock|Power"
} > $O 2>&1
wait $LP
echo "== load samples" >> $O
cat /tmp/load.out >> $O
