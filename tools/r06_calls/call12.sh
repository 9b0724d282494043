#!/bin/bash
# Round 6, call 12: the final file set, part 1 (tools/collect_profiles.sh: default bench under rocprofv3, PMC passes, fp16 table, default line)
bash tools/collect_profiles.sh 2>&1 | tail -25
