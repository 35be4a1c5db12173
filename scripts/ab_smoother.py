"""A/B of library builds on the level-1 passes (dev aid): per build (NDSM_HIP_LIB) the time of the two-sweep, one-sweep and
sweep+residual passes (general / declared-zero rhs) and of a solve-loop cycle at 512^3, plus a checksum of the result
of five sweeps + three cycles - every build must print the same checksum (same bits).
usage: ab_smoother.py <lib.so | default> ..."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
    n = int(os.environ.get("AB_N", "512"))
    out = {}
    mesh = [np.linspace(0, 1, n)] * 3
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
    rng = np.random.default_rng(1)
    u0 = rng.uniform(-1, 1, (n, n, n))
    S.upload(1, _lib.BUF_U, u0); S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
    h = hashlib.sha256()
    for lap in (0, 1):
        if lap:
            S.zero_rhs()
        S.op(_lib.OP_RELAX, 1, 5); S.op(_lib.OP_RELAX_RES, 1, 1)
        h.update(S.download(1, _lib.BUF_U).tobytes()); h.update(S.download(1, _lib.BUF_R).tobytes())
        for name, op, cnt in (("s2", _lib.OP_RELAX, 2), ("s1", _lib.OP_RELAX, 1), ("res", _lib.OP_RELAX_RES, 1)):
            S.op(op, 1, cnt); S.sync()
            out[f"{name}{'L' if lap else 'G'}"] = min(S.timed(lambda: [S.op(op, 1, cnt) for _ in range(5)]) / 5 for _ in range(3)) * 1e3
    S.upload(1, _lib.BUF_U, u0)
    ie, du, nc, hist = S.solve(vc_tol=0.0, nmax=3, hist_len=3)
    h.update(S.download(1, _lib.BUF_U).tobytes()); h.update(np.asarray(hist).tobytes())
    S.solve(vc_tol=0.0, nmax=2); S.sync()
    import time
    t0 = time.perf_counter(); S.solve(vc_tol=0.0, nmax=10); S.sync()
    out["cycleL"] = (time.perf_counter() - t0) / 10 * 1e6
    S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
    S.solve(vc_tol=0.0, nmax=2); S.sync()
    t0 = time.perf_counter(); S.solve(vc_tol=0.0, nmax=10); S.sync()
    out["cycleG"] = (time.perf_counter() - t0) / 10 * 1e6
    out["sha"] = h.hexdigest()[:12]
    S.close()
    print(json.dumps(out))
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["NDSM_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=300)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{os.path.basename(lib):22s}", " ".join(f"{k}:{v:7.0f}" if k != "sha" else f"sha:{v}" for k, v in d.items()), flush=True)
        except Exception:
            print(lib, "FAILED rc", r.returncode, r.stdout[-400:], r.stderr[-600:], flush=True)
