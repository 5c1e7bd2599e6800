#!/bin/bash
# same-box A/B of the host-side switches of rounds 5 and 6 on the composition and inversion steps (graph replays, no roofline leg)
run() { # label, env...
  local label=$1; shift
  for mix in comp inv; do
    env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done
}
for rep in 1 2; do
  run all-on   MVOC_X=1
  run no-tail  MVOC_PRUNE_SOURCE_TAIL=0
  run no-share MVOC_SHARE_CFG_PREFIX=0
  run no-fold  MVOC_GN_FOLD=0
  run no-subpx MVOC_SUBPIXEL=0
  run no-korder MVOC_KORDER=0
  run xs-resid MVOC_XS_RESID_TILED_ROWS=0
  run all-off  MVOC_PRUNE_SOURCE_TAIL=0 MVOC_SHARE_CFG_PREFIX=0 MVOC_GN_FOLD=0 MVOC_SUBPIXEL=0 MVOC_KORDER=0 MVOC_GN_NT_BYTES=99999999999
done
