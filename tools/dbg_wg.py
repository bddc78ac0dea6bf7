import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture
import test_hip_swin_bwd as T
for tag in sys.argv[1:]:
    fx = load_fixture(f"block_{tag}")
    m = fx.meta
    print(tag, m)
    sd = T._rounded_state(fx["sd"])
    res = T._run_block_weight_grads(sd, T.r16(fx["in"]["x"]), fx["in"].get("prompt"), T.r16(fx["in"]["gout"]), m["window"], m["shift"], m["heads"])
    for k, v in res.items():
        print(f"  {k:28s} {v:.4e}")
