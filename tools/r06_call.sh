#!/bin/bash
# Round 6: run one of the recorded gpurun call bodies (tools/r06_calls/callN.sh) on the GPU box:  gpurun -- 'bash tools/r06_call.sh N'
bash "$(dirname "$0")/r06_calls/call$1.sh"
