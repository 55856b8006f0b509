#!/bin/bash
cd "$(dirname "$0")"
for f in pp_test pp_test_*; do [ -x $f ] && [ "$f" != "${f%.hip}.hip" ] && { echo "== $f"; timeout 120 ./$f 2>&1 | grep "pass 1"; }; done
