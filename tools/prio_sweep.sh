# wave-priority flip on/off (HG_PRIO_FLIP) across launch sizes; default = the host's heuristic
for n in 1024 2048 4096 6144 8192 12288 16384; do
  for f in 0 1; do
    HG_PRIO_FLIP=$f timeout -k 10 120 python bench.py --envs-per-gpu $n --steps 200 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('envs $n flip $f', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
