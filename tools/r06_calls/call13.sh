#!/bin/bash
# Round 6, call 13: the final file set, part 2 (tools/collect_configs.sh: BASELINE configs 2 / 4 / 5, B = 8, padded batches, preprocessing,
# the data-parallel code path on one rank in both wire formats)
bash tools/collect_configs.sh 2>&1 | tail -30
