#!/usr/bin/env bash
bash tools/gpu_ab.sh "n0 x1 x2 x3" CartPole-v1 fused,fusedf32,fused-final 1048576 3
