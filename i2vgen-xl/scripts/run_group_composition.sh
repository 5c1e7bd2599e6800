#!/bin/bash
# run from i2vgen-xl/ ; NGPU>1 runs one composition entry per GPU
NGPU=${NGPU:-1}
if [ "$NGPU" -gt 1 ]; then
  PYTHONPATH=.. python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port 29512 \
    composite.py --template_config "configs/group_composite/template.yaml" --configs_json "configs/group_composite/group_config.json" "$@"
else
  PYTHONPATH=.. python composite.py --template_config "configs/group_composite/template.yaml" \
    --configs_json "configs/group_composite/group_config.json" "$@"
fi
