# bench with other builds of the library (tools/libhg_sim_<name>.so, built by hand with extra hipcc flags)
for name in "" ilp clause iterilp iterocc ""; do
  if [ -z "$name" ]; then unset HG_SIM_LIB; label=product; else export HG_SIM_LIB=$(pwd)/tools/libhg_sim_$name.so; label=$name; fi
  timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'finite', d['config']['state_finite'])" || echo "$label failed"
done
