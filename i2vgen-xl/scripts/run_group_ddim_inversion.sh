#!/bin/bash
# run from i2vgen-xl/ ; NGPU>1 shards the group's videos one per GPU (independent processes, no collectives)
NGPU=${NGPU:-1}
if [ "$NGPU" -gt 1 ]; then
  PYTHONPATH=.. python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port 29511 \
    inverse.py --template_config "configs/group_inversion/template.yaml" --configs_json "configs/group_inversion/group_config.json" "$@"
else
  PYTHONPATH=.. python inverse.py --template_config "configs/group_inversion/template.yaml" \
    --configs_json "configs/group_inversion/group_config.json" "$@"
fi
