for v in default bwd1 bwd2; do
  [ $v = default ] && unset INR_LIB_PATH || export INR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_probe/libinr_$v.so
  echo "$v $(python tools/roialign_probe.py 2>&1 | grep -v amdgpu)"
done
