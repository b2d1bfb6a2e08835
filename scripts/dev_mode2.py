"""Per-atom-list mode (k1_mode 2) against the bitmask sub-tile mode (1) on a lone engine in the large-batch decomposition."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from blues_amd import build, integrators, systems
    build.build_engine()
    from blues_amd.engine import NativeEngine
    frozen = sys.argv[2] == "1"
    s, v = systems.s23k(mobile_atoms=275, frozen=frozen)
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=3).to_data(precision=0))
    try:
        f = g.get_forces(); t = g.energy_terms()
    except Exception as e:
        print("ERR", e, g.stats()); raise
    st = g.stats()
    np.save(sys.argv[3], f)
    print(json.dumps({"terms": t.tolist(), "nan": int(np.isnan(f).sum()), "jcap": st["jcap"], "max_jcount": st["max_jcount"], "tiles": st["i_tiles"]}))
    sys.exit(0)
for frozen in ("1", "0"):
    out = {}
    for mode in ("1", "2"):
        env = dict(os.environ, BLUES_FUSE="0", BLUES_SKIN="0.12", BLUES_K2_JITER="4", BLUES_K1_MODE=mode)
        r = subprocess.run([sys.executable, __file__, "child", frozen, "/tmp/f_%s.npy" % mode], env=env, capture_output=True, text=True)
        print("frozen", frozen, "mode", mode, r.stdout.strip()[-600:], r.stderr.strip()[-200:])
        out[mode] = np.load("/tmp/f_%s.npy" % mode)
    d = np.abs(out["1"] - out["2"])
    print("  max |f1 - f2| =", np.nanmax(d), "of", np.abs(out["1"]).max(), "nan atoms:", np.unique(np.nonzero(np.isnan(out["2"]))[0])[:20])
