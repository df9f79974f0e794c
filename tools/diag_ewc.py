import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_parity_gpu import adapter, relmax, relnorm
from oracle import scenarios as sc, fixtures as fx
want = dict(np.load("tests/golden/ewc.npz"))
for dt in ("f32", "bf16"):
    got = sc.scenario_ewc(adapter(dt))
    print(dt, "losses", got["losses"], want["losses"])
    print(" fisher_head_w relmax", relmax(got["fisher_head_w"], want["fisher_head_w"]), "relnorm", relnorm(got["fisher_head_w"], want["fisher_head_w"]))
    print(" fisher_bn1 relnorm", relnorm(got["fisher_bn1"], want["fisher_bn1"]))
    g = dict(zip(got["fisher_names"], got["fisher_rows"])); w = dict(zip(want["fisher_names"], want["fisher_rows"]))
    errs = sorted(((abs(g[n][0]-w[n][0])/max(w[n][1],1e-300), n) for n in w), reverse=True)
    print(" fisher sum errs worst", errs[:4], "median", errs[len(errs)//2])
    g = dict(zip(got["param_names"], got["param_rows"])); w = dict(zip(want["param_names"], want["param_rows"]))
    errs = sorted(((abs(g[n][0]-w[n][0])/max(w[n][1],1e-300), n) for n in w), reverse=True)
    print(" param sum errs worst", errs[:3], "median", errs[len(errs)//2])
    print(" head_w relmax", relmax(got["head_w"], want["head_w"]), "rm_last", relmax(got["rm_last"], want["rm_last"]))
with fx.use_dtype(torch.float32):
    o = sc.scenario_ewc(sc.OracleAdapter())
print("oracle fp32: fisher_head_w relmax", relmax(o["fisher_head_w"], want["fisher_head_w"]), "losses", o["losses"])
