#!/usr/bin/env bash
bash tools/gpu_ab.sh "rE2 rE1 rE3" CartPole-v1 fused,fused-final 1048576 2
