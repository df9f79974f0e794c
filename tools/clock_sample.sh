#!/bin/bash
# usage: tools/clock_sample.sh <tag> <command ...> : runs the command and samples the GPU's shader clock / power / temperature every 0.2 s beside it
# (rocm-smi), printing min / median / max of the samples taken while the command ran.  Evidence for the "the chip holds a lower clock" notes in profiles/.
tag=$1; shift
out=gpurun_out/clock_$tag.txt
mkdir -p gpurun_out
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|Average Graphics|junction" | tr '\n' ' '; echo; sleep 0.2; done ) > $out &
SP=$!
"$@"
kill $SP 2>/dev/null
python3 - "$out" "$tag" <<'P'
import re,sys,statistics as st
sc=[];pw=[]
for l in open(sys.argv[1]):
    m=re.search(r"sclk clock level: \d+: \((\d+)Mhz\)",l)
    if m: sc.append(int(m.group(1)))
    m=re.search(r"Power \(W\): ([0-9.]+)",l)
    if m: pw.append(float(m.group(1)))
def s(v): return "n/a" if not v else f"min {min(v)} median {st.median(v)} max {max(v)} (n={len(v)})"
print(f"[clock {sys.argv[2]}] sclk MHz {s(sc)} | power W {s(pw)}")
P
