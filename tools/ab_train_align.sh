#!/bin/bash
# bench.py's training leg (captured step alone / with the next batch's auction alignment prefetched / with it in series) for a
# list of P2PB_EXPERIMENT settings, alternating on one box:  tools/ab_train_align.sh <rounds> "<setting>" ["<setting>" ...]
cd $GRAFT_REPO_ROOT; n=$1; shift
for i in $(seq $n); do for e in "$@"; do
  P2PB_EXPERIMENT="$e" python bench.py --no-cpu-baseline --no-alt-math --no-pvdl 2>/dev/null | python -c "
import sys, json
t = json.loads(sys.stdin.read().strip().splitlines()[-1])['train_step']
print('experiment=\"$e\": step %.2f ms, with alignment prefetched %.2f ms, in series %.2f ms' % (t['ms_per_step'], t['ms_per_step_with_align'], t['ms_per_step_with_align_serial']))"
done; done
