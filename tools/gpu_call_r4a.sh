#!/usr/bin/env bash
# Round-4 call A: full GPU suite on the exact-band / NaN-clamp build, the Acrobot threshold A/B on the device, dynamic VALU counts,
# timing A/B against the round-3 library, one bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
cp gym_amd/_lib/libmxv.so gym_amd/_lib/variants/libmxv_product.so
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
V=gym_amd/_lib/variants
timeout 300 python tools/acrobot_threshold_ab.py --gpu "round-3 library=$V/libmxv_r3.so" "hot path only (exact band off)=$V/libmxv_band0.so" \
   "exact band off, every cosine direct (own sincos)=$V/libmxv_band0direct.so" "exact band off, ocml sincos, angle addition=$V/libmxv_band0ocml.so" \
   "exact band off, ocml, every cosine direct=$V/libmxv_band0ocmldirect.so" "product (exact band on)=$V/libmxv_product.so" > $O/threshold_ab.jsonl 2> $O/threshold_ab.err
bash tools/gpu_valu.sh r4a product > $O/valu_product.log 2>&1
bash tools/gpu_valu.sh r3 r3 > $O/valu_r3.log 2>&1
bash tools/gpu_ab.sh "r3 product band0 clampcmp" "Acrobot-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0" fused 524288 2 > $O/ab_524288.txt 2>&1
cp gpurun_out/ab.log $O/ab_524288.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/finished
