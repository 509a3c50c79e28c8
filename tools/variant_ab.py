"""A/B of kernel instantiations for one robot: HG_VARIANT=<index> forces a variant (hg_sim.hip, HG_VARIANTS).
usage: python tools/variant_ab.py dancer 65536 4 0 12"""
import os, subprocess, sys
robot, n_envs, sub = sys.argv[1], sys.argv[2], sys.argv[3]
code = f"""
import sys; sys.path.insert(0, '.')
from tools.config_table import run
run('{robot}', {n_envs}, n_substeps={sub}, seed=4)
"""
for v in sys.argv[4:]:
    env = dict(os.environ, HG_VARIANT=v)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"HG_VARIANT={v}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
