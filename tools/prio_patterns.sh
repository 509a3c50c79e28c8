# kernel time of the 4096-env NAO bench for every assignment of the four s_setprio levels to the four wave slots of a SIMD
# (HG_PRIO_FLIP=<hex byte>: bits [2w, 2w+2) = priority of slot w from the solver on; default 0xE4 = 0,1,2,3), plus a few with ties
export HG_BENCH_HANDS=0
for p in e4 e1 d8 d2 c9 c6 b4 b1 9c 93 8d 87 78 72 6c 63 4e 4b 39 36 2d 27 1e 1b  a4 e0 f4 e5 a5 50 fa 55 00 ; do
  if [ $p = 00 ]; then export HG_PRIO_FLIP=0; else export HG_PRIO_FLIP=$p; fi
  timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pattern 0x$p', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'b2b %.4f' % d['roofline']['kernel_ms_back_to_back'])" || exit 1
done
