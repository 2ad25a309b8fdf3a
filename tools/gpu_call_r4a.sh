#!/usr/bin/env bash
# Round-4 call A: full GPU suite on the exact-band / NaN-clamp / trimmed-Acrobot build, the Acrobot threshold A/B on the device, dynamic
# VALU counts, timing A/B against the round-3 library, one bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; rm -rf $O; mkdir -p $O
cp gym_amd/_lib/libmxv.so gym_amd/_lib/variants/libmxv_product.so
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
V=gym_amd/_lib/variants
timeout 300 python tools/acrobot_threshold_ab.py --gpu "round-3 library=$V/libmxv_r3.so" \
   "round-3 arithmetic rebuilt (argument roundings carried, no exact band)=$V/libmxv_band0carry.so" \
   "round-4 hot path alone (plain angle addition, no exact band)=$V/libmxv_band0.so" \
   "no exact band, every cosine direct (own sincos)=$V/libmxv_band0direct.so" \
   "no exact band, ocml sincos, argument roundings carried=$V/libmxv_band0ocmlcarry.so" \
   "no exact band, ocml, every cosine direct=$V/libmxv_band0ocmldirect.so" \
   "product (exact band on)=$V/libmxv_product.so" > $O/threshold_ab.jsonl 2> $O/threshold_ab.err
bash tools/gpu_valu.sh r4b product > $O/valu_product.log 2>&1
bash tools/gpu_ab.sh "r3 product carry nobitop3" "Acrobot-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0" fused 524288 2 > $O/ab_524288.txt 2>&1
cp gpurun_out/ab.log $O/ab_524288.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/finished
