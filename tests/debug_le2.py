import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from systems import *
import test_gpu_le as T
n = 4000
s = T.melted(n, nchains=4, seed=2)
script = T.le_script(n1=5, nl=20, nu=1000, left=1, right=1, lprob="", uprob="", lr="")
tmp = tempfile.mkdtemp()
o = run_oracle(script, s)
p = run_product(script, s, tmp)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 58
done = 0
while done < 58:
    o.run(chunk); p.command("run %d" % chunk); done += chunk
    dx = np.abs(p.gather("x") - o.x()); df = np.abs(p.gather("f") - o.f())
    i = np.unravel_index(dx.argmax(), dx.shape)[0]
    print(done, "max dx %.3e at tag %d  max df %.3e at tag %d  fene warn o=%d p=%d  ext=%d" % (
        dx.max(), i + 1, df.max(), np.unravel_index(df.argmax(), df.shape)[0] + 1, o.fene_warnings(), p.stat("fene_warnings"),
        len([b for b in o.bond_set() if b[0] == 2])))
    if dx.max() > 1e-6:
        nb, bt, ba = o.bond_table()
        print("  bonds of tag", i + 1, ba[i, :nb[i]], "types", bt[i, :nb[i]])
        print("  x_o", o.x()[i], "x_p", p.gather("x")[i], "box", s["box"][0])
        break
