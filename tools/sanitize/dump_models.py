"""Writes <out>/<robot>.bin = bytes(hg_model) + bytes(hg_config) for the sanitizer driver (tools/host_sanitize.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from humanoid_gym_amd.robots import load_robot      # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for name in ("dancer", "nao", "pepper", "romeo", "nao_hands", "romeo_hands"):
    model, cfg = load_robot(name)
    with open(os.path.join(out, name + ".bin"), "wb") as f:
        f.write(bytes(model.to_c()))
        f.write(bytes(cfg.to_c()))
    print(name, "->", os.path.join(out, name + ".bin"))
