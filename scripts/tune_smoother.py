"""Time the level-1 smoother passes for tile configurations NDSM_FUSED_CFG=<s2>,<s1>,<res> (dev aid).
usage: tune_smoother.py 0,0,0 1,3,1 ...   columns: us per launch, general rhs / declared-zero rhs"""
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
        for lap in (0, 1):
            if lap:
                S.zero_rhs()
            for name, op, cnt in (("s2", _lib.OP_RELAX, 2), ("s1", _lib.OP_RELAX, 1), ("res", _lib.OP_RELAX_RES, 1)):
                S.op(op, 1, cnt); S.sync()
                out[f"{n}{name}{lap}"] = min(S.timed(lambda: [S.op(op, 1, cnt) for _ in range(5)]) / 5 for _ in range(3))
        S.close()
    print(json.dumps(out))
else:
    for cfg in sys.argv[1:]:
        env = dict(os.environ, NDSM_FUSED_CFG=cfg)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=200)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("cfg", cfg, " ".join(f"{k}:{float(t)*1e3:6.0f}" for k, t in d.items()), flush=True)
        except Exception as e:
            print("cfg", cfg, "FAILED", r.stdout[-300:], r.stderr[-300:], flush=True)
