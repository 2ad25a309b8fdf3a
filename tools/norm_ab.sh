mkdir -p gpurun_out/r4n; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" noxcd oldred novec; do
  if [ -z "$v" ]; then unset MXV_LIB_PATH; else export MXV_LIB_PATH=$GRAFT_REPO_ROOT/gym_amd/_lib/variants/libmxv_$v.so; fi
  echo "variant=${v:-default} $(python tools/norm_ab.py 2>/dev/null | tail -1)"
done
done
