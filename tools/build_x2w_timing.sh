#!/bin/bash
# tools/exp/lib_x2w.so = the library with every split kernel's low-weight-plane product compiled out (-DP2PB_X2W_TIMING, common.h):
# TIMING ONLY (results are wrong with ordinary packs) -- the time side of profiles/r06_f16x2w_ab.txt. Use: P2PB_LIB_PATH=tools/exp/lib_x2w.so
R=$(cd $(dirname $0)/..; pwd); B=$R/p2p_bridge_amd/csrc/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -DP2PB_X2W_TIMING"
mkdir -p /tmp/x2w $R/tools/exp
( /opt/rocm/bin/hipcc $F -c $R/p2p_bridge_amd/csrc/conv3d.hip -o /tmp/x2w/conv3d.o ) &
( /opt/rocm/bin/hipcc $F -c $R/p2p_bridge_amd/csrc/pointwise.hip -o /tmp/x2w/pointwise.o ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib_x2w.so /tmp/x2w/conv3d.o /tmp/x2w/pointwise.o \
  $(ls $B/*.o | grep -v "/conv3d.o\|/pointwise.o") && echo built lib_x2w
