#!/bin/bash
# tools/exp/lib_<unit><tag>.so = the library with csrc/<unit>.hip recompiled under extra -D switches:
#   tools/build_variant.sh sampling tl "-DFG_TIMELINE"     (use with P2PB_LIB_PATH=tools/exp/lib_samplingtl.so)
R=$(cd $(dirname $0)/..; pwd); B=$R/p2p_bridge_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $3 -c $R/p2p_bridge_amd/csrc/$1.hip -o /tmp/$1_$2.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib_$1$2.so /tmp/$1_$2.o $(ls $B/*.o | grep -v "/$1.o") && echo built $1$2
