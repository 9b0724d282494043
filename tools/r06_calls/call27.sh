# ViLT weight-gradient group beside the LM backward: items per launch sweep, B = 256 (and B = 128), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c27; mkdir -p $O
for it in 0 256 224 192 160 128 0; do
  VAULT_WGRAD_BESIDE_LM=$it timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/b256_$it.json 2> $O/b256_$it.err || { tail -5 $O/b256_$it.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/b256_$it.json").read().strip().splitlines()[-1])
print("beside=$it", "value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], "vilt", d.get("vilt_block_frac"), "lm", d.get("lm_block_frac"))
PY
done
