# A/B of experimental builds of the library (tools/exp/lib_<tag>.so via P2PB_LIB_PATH): convolutions of one evaluation + bench
for rep in 1 2; do
for v in "$@"; do
  export P2PB_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/lib_$v.so
  echo "== $v (rep $rep)"
  timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl 2>&1 | tail -1 | cut -c1-140
done; done
