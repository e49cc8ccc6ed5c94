"""Runs one script on the product engine in a process of its own (test helper of test_gpu_md.py): the step kernel picks
its variant from environment switches that are read once per process.
usage: variant_worker.py SYSTEM.pkl SCRIPT.txt OUT.npz"""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from systems import run_product

system = pickle.load(open(sys.argv[1], "rb"))
script = open(sys.argv[2]).read()
p = run_product(script, system, os.path.dirname(sys.argv[3]))
np.savez(sys.argv[3], x=p.gather("x"), v=p.gather("v"), image=p.gather("image"),
         thermo=np.array([p.get_thermo(k) for k in ("temp", "epair", "emol", "etotal", "press")]),
         builds=np.array([p.stat("neigh_builds")]))
p.close()
