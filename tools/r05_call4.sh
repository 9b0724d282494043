#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "optimizer_leaves or api_backward or tape_replay or two_step or adamw_kernel" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -n "zero mask\|passed\|failed\|Error" $O/tests.log | head
timeout -k 10 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c4/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","ms_per_step_median","step_mfma_frac","vilt_block_frac","lm_block_frac"): print(k, d[k])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("f32pix", d.get("with_f32_pixel_values")); print("h2d", d.get("with_h2d_input_copies",{}).get("value")); print("u8", d.get("with_uint8_input_pipeline",{}).get("value"))
print("fp16", {k:v for k,v in d["parity"]["fp16_operands"].items() if k!="what"})
print("blocks", d["blocks"]["vilt"]["ms_forward"], d["blocks"]["vilt"]["ms_backward"], d["blocks"]["lm"]["ms_forward"], d["blocks"]["lm"]["ms_backward"])
PY
for L in tree gelu_old; do if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi; echo "== $L"; VAULT_HIP_LIB=$P python tools/traj_seeds.py 300 400 500 2>&1 | grep -v amdgpu.ids; done | tee $O/traj_seeds.txt
