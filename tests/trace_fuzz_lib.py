"""The scenario generator of tests/test_gpu_fuzz.py::test_random_le_scenarios for the step tracers."""
import numpy as np
from systems import *
from test_gpu_le import LE, barrier_types, melted

def scenario(seed):
    rng = np.random.RandomState(1000 + seed)
    n = int(rng.choice([1200, 2000, 3500])); nchains = int(rng.choice([1, 2, 5])); frac = float(rng.choice([0.0, 0.1, 0.4]))
    types = barrier_types(n, 50 + seed, frac=frac) if frac > 0 else np.ones(n, dtype=np.int32)
    s = melted(n, nchains=nchains, seed=20 + seed % 3, steps=800, types=types)
    s["ntypes"], s["mass"] = 4, [1.0] * 4
    n1, nl, nu = int(rng.randint(3, 9)), int(rng.randint(4, 11)), int(rng.randint(4, 11))
    tp = float(rng.choice([0.0, 0.3, 0.7, 1.0])); lp, up = float(rng.choice([0.2, 0.6, 1.0])), float(rng.choice([0.1, 0.5, 1.0]))
    lprob = "" if lp >= 1.0 else "prob %g %d" % (lp, 100 + seed); uprob = "" if up >= 1.0 else "prob %g %d" % (up, 200 + seed)
    rmax = float(rng.choice([0.5, 1.3, 2.0])); lr = "4" if rng.rand() < 0.7 else ""
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 8.0 5.0 1.0 1.0")
    script = base + LE.format(n1=n1, nl=nl, nu=nu, neutral=1, left=2, right=3, tp=tp, lr=lr, lprob=lprob, uprob=uprob, rmax=rmax)
    print("seed", seed, dict(n=n, nchains=nchains, frac=frac, n1=n1, nl=nl, nu=nu, tp=tp, lprob=lprob, uprob=uprob, rmax=rmax, lr=lr, L=s["box"][0][1]))
    return s, script

