for wl in gemvtr_f32_65536x4096 gemvtr_f16_65536x4096 gemv_f32_4096x65536 gemv_f16_4096x65536; do
 for round in 1 2; do
  for lib in libwgebra_hip.so libwgebra_hip_gv_w6.so libwgebra_hip_gv_w8.so libwgebra_hip_gv_w16.so; do
    v=$(WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib python bench.py --steps 300 --warmup 30 --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['ms_per_step'])")
    echo "$wl $lib $v"
  done
 done
done
