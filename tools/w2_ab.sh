for L in base nochunkpf; do
  if [ $L = base ]; then unset GEOGRAYPHER_AMD_LIB; else export GEOGRAYPHER_AMD_LIB=$PWD/geograypher_amd/csrc/libgeograster_$L.so; fi
  python bench.py --no-cpu-baseline --no-api --no-io --no-c4 --no-c5 > gpurun_out/w2_$L.json 2>/dev/null
  python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/w2_$L.json") if l.startswith("{")][-1])
w = j["workload_2"]
print("$L", "C2", j["ms_per_step"], j["roofline"]["stage_ms_per_view"]["raster_ms"], "agg", j["aggregate"]["views_per_s"], "| forest", w["scale_1"]["us_per_view"], w["scale_1"]["mpix_per_s"], "| forest q", w["scale_0.25"]["us_per_view"], w["scale_0.25"]["mpix_per_s"])
PY
done
