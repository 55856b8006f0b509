#!/bin/bash
# tools/exp/lib_pw<tag>.so = the library with pointwise.hip recompiled under extra -D switches: tools/build_pw_variant.sh tl "-DPP_TIMELINE"
R=$(cd $(dirname $0)/..; pwd); B=$R/p2p_bridge_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $2 -c $R/p2p_bridge_amd/csrc/pointwise.hip -o /tmp/pointwise_$1.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib_pw$1.so /tmp/pointwise_$1.o $(ls $B/*.o | grep -v "/pointwise.o") && echo built pw$1
