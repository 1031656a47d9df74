out=gpurun_out/trace; mkdir -p $out
for l in DFF_net.dres2.conv6.0 DFF_net.deconv_1.0 DFF_net.dres3.conv1.0.0 DFF_net.dres2.conv0.0.0 DFF_net.dres3.conv0.0.0; do
  DFFW_TRACE_LAYER=$l DFFW_TRACE_OUT=$out/$l.bin python bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 >/dev/null 2>&1
  echo "== $l"
  python tools/trace_report.py $out/$l.bin
done
