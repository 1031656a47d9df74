#!/bin/bash
# Phase timelines of selected layers (needs a library built with `make TRACE=1`); run on the GPU box.
out=${1:-gpurun_out/trace}
mkdir -p $out
for l in DFF_net.dres4.conv0.0.0 DFF_net.dres4.conv6.0 DFF_net.FM_measure.Focus_extraction.2.Focus_Measure.conv.2.0 DFF_net.dres4.conv1.0.0 DFF_net.dres2.conv0.0.0 DFF_net.FM_measure.Focus_extraction.0.0; do
  DFFW_TRACE_LAYER=$l DFFW_TRACE_OUT=$out/$l.bin python bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 >/dev/null 2>&1
  echo "== $l"
  python tools/trace_report.py $out/$l.bin
done
