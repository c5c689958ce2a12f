#!/bin/bash
# Is the chip at its power limit under the two headline kernels?  Samples rocm-smi (power, clocks) every 0.25 s while
# bench.py runs 60 steps of the north-star configuration, then the sustained bare-MFMA probe for comparison.
out=${1:-gpurun_out/power}
mkdir -p $out
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showperflevel --json 2>/dev/null | head -c 2000; echo; sleep 0.25; done ) > $out/smi_bench.log &
SMI=$!
python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs > $out/bench.log 2>&1
sleep 1
echo "=== PROBE ===" >> $out/smi_bench.log
[ -x tools/mfma_sustained_probe.bin ] && ./tools/mfma_sustained_probe.bin 20000 8 > $out/probe.log 2>&1
kill $SMI 2>/dev/null
rocm-smi --showmaxpower --showpowercap 2>/dev/null | tail -12 > $out/caps.log
python3 - $out <<'PY'
import json, sys, re
out = sys.argv[1]
rows = []
phase = "bench"
for ln in open(out + "/smi_bench.log"):
    if ln.startswith("=== PROBE"):
        phase = "probe"; continue
    try:
        d = json.loads(ln)
    except Exception:
        continue
    c = d.get("card0", {})
    p = [v for k, v in c.items() if "ower" in k and "W" in k]
    s = [v for k, v in c.items() if k.startswith("sclk")]
    rows.append((phase, p[:1], s[:1]))
for ph in ("bench", "probe"):
    r = [x for x in rows if x[0] == ph]
    print(ph, len(r), "samples; power:", sorted(set(str(x[1]) for x in r))[-6:], "sclk:", sorted(set(str(x[2]) for x in r))[-6:])
print(open(out + "/caps.log").read())
PY
