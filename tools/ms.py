"""Print ms_per_step (and full-step ms) from a bench.py output file (last JSON line)."""
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith("{")]
d = json.loads(lines[-1])
print(sys.argv[2] if len(sys.argv) > 2 else "", d["ms_per_step"], d.get("full_step_ms_with_clip_adam"))
