"""Same-box A/B of the step kernel across source trees: each directory under ab/ holds a `git archive` of one commit
(humanoid_gym_amd + include + tools/config_table.py) with its own built libhg_sim.so; the current tree runs last.
usage (GPU box): python tools/ab_commits.py dancer 65536 4 [cfg_key=value ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
robot, n_envs, sub = sys.argv[1], sys.argv[2], sys.argv[3]
kw = ", ".join(sys.argv[4:])
trees = sorted(os.path.join(ROOT, "ab", d) for d in os.listdir(os.path.join(ROOT, "ab"))) + [ROOT]
for rep in range(2):
    for t in trees:
        code = (f"import sys; sys.path.insert(0, {t!r}); sys.path.insert(0, {os.path.join(t, 'tools')!r}); "
                f"from config_table import run; run({robot!r}, {n_envs}, n_substeps={sub}, seed=4{', ' + kw if kw else ''})")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=t)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        print(f"{os.path.basename(t):10s} {line}", flush=True)
