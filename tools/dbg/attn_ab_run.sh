#!/bin/bash
# GPU box: attention bench per A/B library (tools/dbg/attn_ab.sh builds them); a tag "x:ENV=VAL" sets an environment variable too
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  echo "== $spec"
  export MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_attn_$tag.so
  env $envs python tools/attn_bench.py 5 2>&1 | grep -E "^(self|pair)"
  env $envs python tools/attn_bench.py 1 2>&1 | grep -E "^(self|pair)" | head -4
done
