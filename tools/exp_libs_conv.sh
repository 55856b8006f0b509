# per-instance conv timing of experimental library builds (timing-only ablations produce wrong values)
for v in "$@"; do
  export P2PB_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so
  echo "== $v"
  timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -17 | cut -d, -f1-7
done
