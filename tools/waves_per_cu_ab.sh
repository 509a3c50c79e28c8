for k in 0 8 4; do
  echo "== HG_WAVES_PER_CU=$k"
  HG_WAVES_PER_CU=$k timeout -k 10 120 python -c "
import sys; sys.path.insert(0,'tools')
from config_table import run
run('nao_hands',4096); run('romeo',8192); run('romeo_hands',8192)
" 2>&1 | grep envs || exit 1
  HG_WAVES_PER_CU=$k timeout -k 10 120 python bench.py --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])" || exit 1
done
