#!/bin/bash
# tools/power_sample.sh LABEL -- CMD...: runs CMD while sampling the GPU's socket power and clocks (rocm-smi, every ~50 ms) into
# gpurun_out/power_LABEL.txt: what the card draws and what it clocks at under the single-query kernel, the query-pair kernel and a row
# microbenchmark (round 6: what holds the single-query kernel's clock at 2.2 GHz?)
L=$1; shift; [ "$1" = "--" ] && shift
O=gpurun_out/power_$L.txt; mkdir -p gpurun_out
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.05; done ) > $O.raw 2>/dev/null &
S=$!
"$@" > gpurun_out/power_$L.cmd.log 2>&1; RC=$?
kill $S 2>/dev/null; wait $S 2>/dev/null
python3 - "$O.raw" > $O <<'PY'
import json, sys, re
pw, sc = [], []
for line in open(sys.argv[1]):
    try: d = json.loads(line)
    except ValueError: continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "Power" in k and "W" in k:
            try: pw.append(float(v))
            except ValueError: pass
        if k.startswith("sclk clock level") or k == "sclk clock speed:":
            m = re.search(r"(\d+)Mhz", str(v))
            if m: sc.append(int(m.group(1)))
if pw:
    pw.sort(); print(f"power samples {len(pw)}: median {pw[len(pw)//2]:.0f} W, p90 {pw[int(len(pw)*0.9)]:.0f} W, max {pw[-1]:.0f} W")
if sc:
    sc.sort(); print(f"sclk samples {len(sc)}: median {sc[len(sc)//2]} MHz, min {sc[0]} MHz, max {sc[-1]} MHz")
if not pw and not sc: print("no samples parsed; first raw line:", open(sys.argv[1]).readline()[:400])
PY
rm -f $O.raw; echo "rc=$RC"; cat $O
