#!/bin/bash
# tools/exp/lib_<tag>.so = the library with conv3d.hip recompiled under extra -D switches (timing ablations / variants);
# usage: tools/build_conv_variants.sh tag1:"-DX=1 -DY" tag2:"-DZ" ...   (built in parallel; run via P2PB_LIB_PATH)
R=$(cd $(dirname $0)/..; pwd)
B=$R/p2p_bridge_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
mkdir -p /tmp/convvar $R/tools/exp
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $R/p2p_bridge_amd/csrc/conv3d.hip -o /tmp/convvar/conv3d_$tag.o 2> /tmp/convvar/$tag.err &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib_$tag.so /tmp/convvar/conv3d_$tag.o $(ls $B/*.o | grep -v "/conv3d.o") &&
    echo "built $tag" || { echo "FAILED $tag"; tail -5 /tmp/convvar/$tag.err; } ) &
done
wait
