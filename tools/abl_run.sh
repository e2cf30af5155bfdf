# ablation of k_trigemm_sq's main loop: variants built with -DBOHIP_ABL=x under csrc/abl/ (results are wrong by design)
cd bayesianoptimization.jl_amd/csrc
cp libbohip.so /tmp/libbohip_real.so
for a in real 1 3 7 2; do
  if [ $a = real ]; then cp /tmp/libbohip_real.so libbohip.so; else cp abl/libbohip_abl$a.so libbohip.so; fi
  (cd ../..; echo "ABL=$a: $(python bench.py --no-cpu-baseline --steps 50 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms']['trigemm_sq'],4))")")
done
cp /tmp/libbohip_real.so libbohip.so
