# A/B of the wave-priority flip on the bench workload: which envs of a launch flip (envs >= HG_FLIP_FROM).
# usage (GPU box): bash tools/prio_ab.sh
for k in "" 0 1024 2048 2304 2560 2944 3072 3584 4096; do
  echo "== HG_FLIP_FROM=$k"
  if [ -z "$k" ]; then unset HG_FLIP_FROM; else export HG_FLIP_FROM=$k; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('nao_without_finger_joints',{}).get('ms_per_step'))" || exit 1
done
