#!/bin/bash
# builds the standalone harness against the in-tree library (container; hipcc cross-compiles)
set -e
cd "$(dirname "$0")/../../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics $PPFLAGS -Wno-unused-value \
  tools/exp/pp/pp_test.hip -o tools/exp/pp/pp_test$PPTAG -Lp2p_bridge_amd -lp2pb_hip -Wl,-rpath,'$ORIGIN/../../../p2p_bridge_amd' \
  -Rpass-analysis=kernel-resource-usage 2> tools/exp/pp/build.log || { tail -30 tools/exp/pp/build.log; exit 1; }
grep -A9 "pw_pingpong" tools/exp/pp/build.log | grep -E "Name|VGPRs|Spill|Scratch|Occupancy|SGPRs:" | head -20
