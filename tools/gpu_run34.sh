#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== torchrun 1 rank"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
echo "=== sharded world=1 gather path"; timeout 300 python - <<'PY'
import sys, os
sys.path.insert(0, '.')
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29518", RANK="0", WORLD_SIZE="1", TORCH_NCCL_HIGH_PRIORITY="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from gym_amd.distributed import ShardedRollout
sr = ShardedRollout("CartPole-v1", 1 << 18, seed=0, action_seed=1)
sr.reset(seed=0)
traj = sr.engine.trajectory_buffers(64)
for i in range(4):
    sr.rollout_per_step(64, out=traj)
    sr.gather_async()
g = sr.gather()
print("gathered", [tuple(t.shape) for t in g], float(g.reward.sum()))
rn = sr.make_normalizer()
y = rn.normalize_obs(traj["obs"]); o = rn.normalize_rewards(traj["reward"], traj["terminated"], traj["truncated"])
sr.synchronize(); torch.cuda.synchronize()
print("normalizer over nccl world=1:", tuple(y.shape), float(y.mean()), rn.obs_rms.count)
sr.close(); dist.barrier(); dist.destroy_process_group()
PY
} > gpurun_out/run34.log 2>&1
tail -c 2500 gpurun_out/run34.log
