#!/bin/bash
# dev: A/B of option search_fused on the headline command (same box, alternating)
cd "$(dirname "$0")/.."
for f in ${AB_ORDER:-0 1 0 1}; do
  python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-secondary --no-cpu-baseline --no-pipelined --fused $f 2>/dev/null > /tmp/ab_$f.json
  python - $f <<'PY'
import json, sys
f = sys.argv[1]
o = json.loads([l for l in open(f"/tmp/ab_{f}.json") if l.startswith("{")][-1])
u = o.get("unramped_contract_region") or {}
print("fused", f, "ms/step %.2f us" % (o["ms_per_step"] * 1e3), "steady %.2f" % (o["steady_state_400_steps"]["ms_per_step"] * 1e3),
      "unramped %.2f" % (u.get("ms_per_step", 0) * 1e3), "kernels", {k: round(v * 1e3, 2) for k, v in o["kernels_ms"].items()},
      "span %.2f" % (o["roofline"]["kernel_ms_in_kernel_span"] * 1e3), "parity", o["parity"]["ids_equal_float64_oracle"])
PY
done
