#!/bin/bash
# Samples rocm-smi (power, shader / memory clocks) every 0.25 s while the bench command runs: evidence for the power-limited claim
# of DESIGN.md.  usage: bash tools/power_trace.sh <out.txt> [bench args]
OUT=$1; shift
( while true; do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Max" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT.raw &
S=$!
python bench.py --no-secondary --no-cpu-baseline --steps 60 --warmup 5 "$@" > $OUT.bench.json 2>/dev/null
kill $S
python - <<PY
import re, json
rows = [l for l in open("$OUT.raw") if "Power" in l]
pw = [float(x) for l in rows for x in re.findall(r"Power \(W\): ([0-9.]+)", l)]
pw = pw or [float(x) for l in rows for x in re.findall(r"Average Graphics Package Power \(W\): ([0-9.]+)", l)]
sc = [int(x) for l in rows for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", l)]
mx = re.findall(r"Max Graphics Package Power \(W\): ([0-9.]+)", "".join(rows))
d = json.load(open("$OUT.bench.json"))
with open("$OUT", "w") as f:
    f.write("bench: %s images/s, %s ms per step, dominant launch %s ms\n" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
    f.write("samples: %d; power W: min %s max %s mean %s; max package power: %s\n" % (len(pw), min(pw) if pw else None, max(pw) if pw else None, round(sum(pw) / max(len(pw), 1), 1), mx[:1]))
    f.write("sclk MHz under load (samples above 1000): %s\n" % sorted(set(x for x in sc if x > 1000)))
    f.write("first raw line: " + (rows[0] if rows else "none"))
    f.write("a raw line under load: " + (rows[len(rows) // 2] if rows else "none"))
print(open("$OUT").read())
PY
