#!/bin/bash
# Scaling sweep on one node: `python bench.py --gpus N` (the script launches its N ranks itself: one process per GPU,
# RCCL, rendezvous on 127.0.0.1) for N = 1, 2, 4, 8 on the headline workload (hero_cfg3, BASELINE.json configs[2]) and the
# sharded keyframe stream (hero_cfg4_stream, configs[3]); prints frames/s and the weak-scaling efficiency
# value(N) / (N * value(1)).   usage: scripts/scale_sweep.sh [steps] [warmup] [Ns...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
STEPS=${1:-20}; WARM=${2:-3}; shift 2 2>/dev/null
NS=${@:-1 2 4 8}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
HAVE=$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
O=${SR_SWEEP_OUT:-gpurun_out}; mkdir -p "$O"
for wl in hero_cfg3 hero_cfg4_stream; do
  base=""
  for n in $NS; do
    if [ "$n" -gt "$HAVE" ]; then echo "$wl N=$n: skipped ($HAVE GPU(s) visible)"; continue; fi
    extra="--no-cpu-baseline --no-roofline"
    line=$(timeout 1200 python bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --workload "$wl" $extra 2> "$O/scale_${wl}_$n.err" | grep '^{' | tail -1)
    echo "$line" > "$O/scale_${wl}_$n.json"
    python - "$wl" "$n" "$base" <<'PY' "$line"
import json, sys
wl, n, base, line = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
try:
    d = json.loads(line)
except Exception:
    print(f"{wl} N={n}: no bench line (see the .err file)"); sys.exit(0)
v = d["value"]
eff = f"  efficiency {v / (n * float(base)):.3f}" if base else ""
print(f"{wl} N={n}: {v:9.1f} frames/s  {d['ms_per_step']:7.2f} ms/step{eff}")
PY
    if [ -z "$base" ] && [ "$n" = "1" ]; then base=$(python -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$line" 2>/dev/null); fi
  done
done
