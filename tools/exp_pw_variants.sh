#!/bin/bash
# timing-only ablations of pw_split_kernel on the 512 -> 1024 launch: variant libraries built with -DPWS_EXP_* (wrong
# numbers, timing only), selected through P2PB_LIB_PATH. Build here (container), run on the GPU box.
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value"
mkdir -p tools/exp/var
for v in ${VARIANTS:-NODMA NOXF NOSPLIT NOMFMA "NOXF -DPWS_EXP_NOSPLIT" "NOXF -DPWS_EXP_NOSPLIT -DPWS_EXP_NODMA"}; do
  name=$(echo $v | tr -d ' ' | sed 's/-DPWS_EXP_/_/g')
  /opt/rocm/bin/hipcc $FLAGS -DPWS_EXP_$v -c p2p_bridge_amd/csrc/pointwise.hip -o tools/exp/var/pw_$name.o 2>/dev/null
  objs=$(ls p2p_bridge_amd/csrc/build/*.o | grep -v pointwise.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/var/lib_$name.so $objs tools/exp/var/pw_$name.o
  echo built $name
done
