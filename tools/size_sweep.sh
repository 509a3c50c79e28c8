for n in 2048 4096 6144 8192 16384 32768; do
    timeout -k 10 120 python bench.py --envs-per-gpu $n --steps 200 --warmup 30 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nao envs $n', 'value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])" || exit 1
done
