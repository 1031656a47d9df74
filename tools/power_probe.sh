#!/bin/bash
# Samples socket power, the XCD clocks and the hot-spot temperature (amd-smi) about five times a second while a command runs.
# usage: tools/power_probe.sh <out.txt> <command...>      one line per sample: power_W clk0..clk7_MHz hotspot_C
out=$1; shift
( while true; do
    amd-smi metric -g 0 --power --clock --temperature 2>/dev/null | awk '
      /SOCKET_POWER/ {p=$2} /GFX_[0-7]:/ {g=1} g && /^ *CLK:/ {c=c" "$2; g=0} /HOTSPOT/ {h=$2} END {print p, c, h}'
  done ) > $out 2>&1 &
spid=$!
"$@"
rc=$?
kill $spid 2>/dev/null
exit $rc
