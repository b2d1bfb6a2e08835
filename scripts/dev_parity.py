#!/usr/bin/env python3
"""Developer check: GPU engine vs CPU oracle on the 975-atom toluene box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
from oracle import oracle

def rel(a, b):
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())

def main():
    s, v = systems.toluene_box()
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.004, temperature=300.0, seed=7)
    for prec in (1, 0):
        idata = integ.to_data(precision=prec)
        o = oracle.Oracle(s, idata); o.set_velocities(v)
        g = NativeEngine(s, idata); g.set_velocities(v)
        print("== precision", "double" if prec else "mixed", g.stats())
        for (ls, le) in ((1.0, 1.0), (0.5, 0.3), (0.0, 0.0)):
            eo, fo, to = o.energy_forces(ls, le)
            g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
            tg = g.energy_terms(); fg = g.get_forces()
            print(" lambda", ls, le, "E oracle %.6f gpu %.6f" % (eo, tg.sum()))
            for k, name in enumerate(("bond","angle","tors","nb","exc","alchS","alchE","restr")):
                if to[k] != 0 or tg[k] != 0:
                    print("   %-6s oracle % .8e gpu % .8e rel %.2e" % (name, to[k], tg[k], abs(tg[k]-to[k])/max(1e-30,abs(to[k]))))
            print("   forces: max|dF|/max|F| = %.3e  rms rel = %.3e" % (rel(fg, fo), np.linalg.norm(fg-fo)/np.linalg.norm(fo)))
        g.set_global("lambda_sterics", 1.0); g.set_global("lambda_electrostatics", 1.0)
        t0 = time.time(); wg = g.run_switch(nsteps, trace=True); tg_ = time.time() - t0
        wo = []
        t0 = time.time()
        for k in range(nsteps):
            o.step(1); wo.append(o.get_global("protocol_work"))
        to_ = time.time() - t0
        wo = np.array(wo)
        print(" work oracle", wo[[0, nsteps//2, -1]], "gpu", wg[[0, nsteps//2, -1]])
        print(" work trace max abs diff %.3e  rel(final) %.3e ; t_gpu %.3fs t_cpu %.3fs" % (np.abs(wg-wo).max(), abs(wg[-1]-wo[-1])/max(1e-30,abs(wo[-1])), tg_, to_))
        print(" pos diff %.3e vel diff %.3e" % (np.abs(g.get_positions()-o.get_positions()).max(), np.abs(g.get_velocities()-o.get_velocities()).max()))
        print(" stats", g.stats(), "oracle evals", o.num_evaluations())
        print(" K1 usec/launch", g.time_nonbonded(20))
        g.close()

if __name__ == "__main__":
    main()
