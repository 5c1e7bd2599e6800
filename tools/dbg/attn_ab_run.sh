#!/bin/bash
# GPU box: attention bench + flash tests per A/B library (tools/dbg/attn_ab.sh builds them)
for tag in "$@"; do
  echo "== $tag"
  export MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_attn_$tag.so
  python tools/attn_bench.py 5 2>&1 | grep -E "^(self|pair)"
  python tools/attn_bench.py 1 2>&1 | grep -E "^(self|pair)" | head -4
done
