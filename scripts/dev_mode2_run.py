"""Runs R chains for N steps in mode 1 and in mode 2 (same seeds) and reports where the work traces part."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, N = int(os.environ.get("RR", "8")), int(os.environ.get("NN", "300"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from blues_amd import build, integrators, systems
    build.build_engine()
    from blues_amd.engine import NativeEngine, NativeBatch
    from blues_amd.replicas import replica_seed
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    engs = []
    for r in range(R):
        g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=N, dt=0.004, temperature=300.0, seed=replica_seed(1234, r)).to_data(precision=0, replica=r))
        g.set_velocities(v); engs.append(g)
    B = NativeBatch(engs)
    out = []
    try:
        for blk in range(N // 10):
            errs, w = B.step(10, trace=True, raise_errors=False)
            out.append(w.tolist())
            bad = [r for r, e in enumerate(errs) if e is not None]
            if bad:
                print("ERR at block", blk, "members", bad, str(errs[bad[0]]), file=sys.stderr); break
    finally:
        st = engs[0].stats()
        print("stats", st, file=sys.stderr)
    np.save(sys.argv[2], np.concatenate([np.array(o) for o in out], axis=1))
    sys.exit(0)
res = {}
for mode in ("1", "2"):
    env = dict(os.environ, BLUES_FUSE="0", BLUES_SKIN="0.12", BLUES_K2_JITER="4", BLUES_K1_MODE=mode)
    r = subprocess.run([sys.executable, __file__, "child", "/tmp/w_%s.npy" % mode], env=env, capture_output=True, text=True)
    print("mode", mode, r.stderr.strip()[-700:])
    res[mode] = np.load("/tmp/w_%s.npy" % mode)
n = min(res["1"].shape[1], res["2"].shape[1])
d = np.abs(res["1"][:, :n] - res["2"][:, :n])
for r in range(R):
    big = np.nonzero(d[r] > 1e-3 * max(1.0, np.abs(res["1"][r, :n]).max()))[0]
    print("member", r, "steps", n, "max diff %.3e" % d[r].max(), "first step off by >1e-3:", big[:1], "w1 %.4f w2 %.4f" % (res["1"][r, n - 1], res["2"][r, n - 1]))
