import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import gym_amd
mode = int(sys.argv[1])
n = [8, 1024, 70000, 300000][mode % 4]
env = gym_amd.make("CartPole-v1", num_envs=n, **({} if mode % 3 else dict(copy=False)))
env.reset(seed=mode)
env.action_space.seed(mode)
keep = []
for i in range(40):
    out = env.step(env.action_space.sample())
    if i % 7 == 0:
        keep.append(out)
if mode % 2:
    env.close()
print("ok", mode, n, float(keep[-1][1].sum()))
# arrays (views of pinned blocks) and possibly the env itself are still alive at interpreter exit
