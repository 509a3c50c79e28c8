# per-slot priority patterns at solver entry (HG_PRIO_FLIP=<hex byte>, 2 bits per wave slot), NAO 4096 envs
for f in e4 d2 c6 d8 e1 b1 78 93 e4; do
    HG_PRIO_FLIP=$f timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pattern $f', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])" || exit 1
done
