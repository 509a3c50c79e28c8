for cfg in "16 1" "16 0" "64 1" "64 0"; do set -- $cfg
  HG_STATS_EVERY=$1 HG_STATS_SIDE=$2 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --stream-groups 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('every $1 side $2', 'value %.2f M' % (d['value']/1e6), 'ms %.4f' % d['ms_per_step'])" || exit 1
done
HG_STATS_EVERY=16 timeout -k 10 200 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no dist every 16', 'value %.2f M' % (d['value']/1e6), 'ms %.4f' % d['ms_per_step'])"
