# Same-box A/B: link records in registers + readlane (product) against the LDS reads (tools/libhg_sim_norec.so, -DHG_RECREG=0).
# Build the comparison library first (CPU, one minute):  python -c "from humanoid_gym_amd.build import build; import os; build(lib=os.path.abspath('tools/libhg_sim_norec.so'), extra_flags=['-DHG_RECREG=0'], obj_dir='_obj_norec')"
for rep in 1 2; do
for name in "" norec; do
  if [ -z "$name" ]; then unset HG_SIM_LIB; label=product; else export HG_SIM_LIB=$(pwd)/tools/libhg_sim_$name.so; label=$name; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --stream-groups 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'bench value %.2f M' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])" || echo "$label failed"
  timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'tools')
from config_table import run
run('nao_hands',4096); run('romeo',8192); run('romeo_hands',8192)" 2>&1 | grep envs | sed "s/^/$label /"
done
done
