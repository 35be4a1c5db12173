"""Time the level-1 smoother for the tile configurations selected by NDSM_FUSED_CFG (dev aid)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
    out = {}
    for n in (512, 256):
        mesh = [np.linspace(0, 1, n)] * 3
        S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
        rng = np.random.default_rng(1)
        S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n))); S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
        S.op(_lib.OP_RELAX, 1, 3); S.sync()
        ts = [S.timed(lambda: S.op(_lib.OP_RELAX, 1, 10)) / 10 for _ in range(3)]
        out[n] = min(ts)
        S.close()
    print(json.dumps(out))
else:
    for cfg in sys.argv[1:]:
        env = dict(os.environ, NDSM_FUSED_CFG=cfg)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=200)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("cfg", cfg, " ".join(f"{n}^3: {float(t)*1e3:7.1f} us = {24*int(n)**3/float(t)/1e6/1e3:6.2f} TB/s" for n, t in d.items()), flush=True)
        except Exception as e:
            print("cfg", cfg, "FAILED", r.stdout[-300:], r.stderr[-300:], flush=True)
