#!/bin/bash
# tools/exp/lib_<tag>.so = the library with ALL THREE translation units of conv3d.hip recompiled under extra -D switches
# (f16x3 forms, bf16x6 forms, bf16x3 data-gradient forms): tools/build_conv_variant_all.sh tag "-DX=1"
R=$(cd $(dirname $0)/..; pwd); B=$R/p2p_bridge_amd/csrc/build; tag=$1; defs=$2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
mkdir -p /tmp/convvar $R/tools/exp
/opt/rocm/bin/hipcc $FLAGS $defs -c $R/p2p_bridge_amd/csrc/conv3d.hip -o /tmp/convvar/c0_$tag.o &
/opt/rocm/bin/hipcc $FLAGS $defs -DCONV_TU=6 -c $R/p2p_bridge_amd/csrc/conv3d.hip -o /tmp/convvar/c6_$tag.o &
/opt/rocm/bin/hipcc $FLAGS $defs -DCONV_TU=3 -c $R/p2p_bridge_amd/csrc/conv3d.hip -o /tmp/convvar/c3_$tag.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib_$tag.so /tmp/convvar/c0_$tag.o /tmp/convvar/c6_$tag.o /tmp/convvar/c3_$tag.o $(ls $B/*.o | grep -v "/conv3d") && echo built $tag
