# wave-priority flip: default heuristic vs never (HG_PRIO_FLIP=0) vs every env (HG_PRIO_FLIP=1), across launch sizes
for n in 2048 4096 6144 8192 16384 32768; do
  for f in default 0 1; do
    if [ $f = default ]; then unset HG_PRIO_FLIP; else export HG_PRIO_FLIP=$f; fi
    timeout -k 10 120 python bench.py --envs-per-gpu $n --steps 200 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nao envs $n flip $f', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
