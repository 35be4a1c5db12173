#!/usr/bin/env python3
"""bench.py - the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 512]

Workload (BASELINE.json: "fine-grid LUP/s + V-cycles/sec at 512^3 fp64"):
config[2], the 512^3 vector-potential problem on ONE GPU - one STEP is one
multigrid V-cycle (ms = 5, reference level rule: 8 grids) of its Ax Laplace
component, boundary data from the analytic field of the reference's
integration test, arrays resident in HBM before the timed region starts.

value = fine-grid lattice-point updates per second over the WHOLE V-cycle:
        2*ms*nx*ny*nz smoother updates on level 1 per cycle / time per cycle
        (coarse levels, residual, transfers, the convergence metric of update_u
        and the host's read-back of it are all inside the timed region and count
        as overhead, not as updates).
Extra keys: vcycles_per_s, smoother (kernel-only, HIP events on the library
stream), roofline = the kernel that dominates the timed region (two-sweep
Laplace launch of level 1): achieved / frac = HBM bytes the launch really moved
(rocprofv3 counters, measured in this run) / launch time / 8 TB/s, with the
algorithmic figure (16 B/LUP x 2 sweeps) beside it as frac_algorithmic;
roofline_general_rhs (the same launch with a right-hand side in HBM), solve
(whole Ax solve to vc_tol=1e-10), configs (BASELINE configs 0, 1, 3 on this GPU),
end_to_end / end_to_end_small (ndsm_vector_solve wall time, host buffers in and
out), cpu_baseline (the reference itself on this box's host cores, same grid).

One process per GPU.  `python3 bench.py --gpus N` without a launcher starts its
own N ranks (torch.distributed.run as a child, before anything touches a GPU)
and relays rank 0's line; under a launcher (WORLD_SIZE set) it is one rank.
N > 1: see DESIGN.md section "multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_LUP = 24.0       # u read + rhs read + u write per full red+black sweep


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_child(cmd, timeout_s, **kw):
    """subprocess.run in a process group of its own; whatever the child leaves behind (profiler helpers,
    launcher agents) is ended with the group, so that nothing this bench started outlives it"""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
        return 124, out, err
    finally:
        try:
            os.killpg(p.pid, signal.SIGTERM)      # the group leader has exited: only stragglers are left
        except OSError:
            pass
    return p.returncode, out, err


def spawn_ranks(args):
    """`python3 bench.py --gpus N` as a lone process: start the N ranks (one per GPU) under torch.distributed.run
    as a CHILD - this process has not loaded the library or touched a GPU - relay rank 0's JSON line, return
    the launcher's exit code (non-zero if any rank failed)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rc, out, err = run_child(cmd, 3000, env=env, cwd=ROOT)
    sys.stderr.write(err[-20000:])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 3
    return rc


EXIT_OVERLAP_HUNG = 17     # a worker's way of saying: the overlapped exchange never came back - start me again without it


def supervise_rank():
    """A rank of an N > 1 job does its work in a CHILD process and only watches it.  The overlapped halo
    exchange (one RCCL communicator driven from two streams) has met real multi-GPU hardware in no run this
    repository could make; should it wedge, a process that already holds the GPU cannot fall back by itself -
    its streams are stuck behind the collective.  The worker's watchdog ends it with EXIT_OVERLAP_HUNG instead,
    every rank's supervisor sees that, and all of them start a second worker with NDSM_HIP_OVERLAP=0 (exchange
    on the main stream) on a rendezvous of its own.  This process never loads the library or touches a GPU."""
    import subprocess
    rc = 3
    for attempt in (0, 1):
        env = dict(os.environ, NDSM_BENCH_WORKER="1", NDSM_BENCH_ATTEMPT=str(attempt))
        if attempt == 1:
            env["NDSM_HIP_OVERLAP"] = "0"
            # (its rendezvous is a store rank 0's worker hosts: with this set, torch would look for the launcher's)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)
        if rc != EXIT_OVERLAP_HUNG:
            break
        print(f"bench.py[rank {os.environ.get('RANK', '?')}]: the worker gave up on the overlapped exchange" +
              (": starting it again with NDSM_HIP_OVERLAP=0" if attempt == 0 else " twice"), file=sys.stderr, flush=True)
    return rc


class Watchdog:
    """ends THIS process with `code` unless cancelled within `seconds` (a wedged collective cannot be interrupted)"""

    def __init__(self, seconds, code, what):
        import threading

        def fire():
            print(f"bench.py[rank {os.environ.get('RANK', '?')}]: {what} did not finish within {seconds:.0f} s - "
                  f"exit {code}", file=sys.stderr, flush=True)
            os._exit(code)
        self.t = threading.Timer(seconds, fire)
        self.t.daemon = True
        self.t.start()

    def cancel(self):
        self.t.cancel()


def boundary_problem(n):
    """Ax of the analytic field (integration_test1.py:57-99) on the four
    Dirichlet faces of the Ax problem ("NDDNDD"), zero elsewhere, rhs = 0."""
    x = np.linspace(0.0, 1.0, n)
    dx = x[1] - x[0]
    y = np.arange(n) * dx
    z = np.arange(n) * dx
    wn = np.pi
    l = np.sqrt(2 * wn ** 2)
    u = np.zeros((n, n, n))                       # (nz, ny, nx)
    ax = lambda X, Y, Z: -np.cos(wn * X) * np.sin(wn * Y) * np.exp(-l * Z)  # noqa: E731
    Zg, Xg = np.meshgrid(z, x, indexing="ij")
    u[:, 0, :] = ax(Xg, y[0], Zg)
    u[:, -1, :] = ax(Xg, y[-1], Zg)
    Yg, Xg = np.meshgrid(y, x, indexing="ij")
    u[0, :, :] = ax(Xg, Yg, z[0])
    u[-1, :, :] = ax(Xg, Yg, z[-1])
    return [x, y, z], u


def cpu_baseline(n=512, seconds_budget=8.0):
    """The reference itself (oracle/_ref: red_black_gauss_3D, ndsm_optimized.f90:40, and solve_poisson_bvp,
    ndsm_poisson.f90:63) on this box's host cores, ON THE BENCHMARKED GRID (n^3); falls back to the C
    port if oracle/_ref is absent.  A bounded sample: ~seconds_budget of smoother sweeps + ONE pass of the
    reference's solve loop (V-cycle + update_u)."""
    from oracle import Oracle, have_ref, usable_cpus
    kind = "reference" if have_ref() else "port"
    orc = Oracle("ref" if have_ref() else "port")
    mesh, u0 = boundary_problem(n)
    rng = np.random.default_rng(2112)
    u = rng.uniform(-1, 1, (n, n, n))
    rhs = np.random.default_rng(2113).uniform(-1, 1, (n, n, n))
    u = orc.relax3d(u, rhs, mesh, "NDDNDD", inplace=True)       # warm-up (thread team, page faults)
    sweeps, t0 = 0, time.perf_counter()
    while True:
        orc.relax3d(u, rhs, mesh, "NDDNDD", inplace=True)      # in place, as the reference's solver calls it
        sweeps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or sweeps >= 2000:
            break
    per_sweep = el / sweeps
    out = {"value": n ** 3 / per_sweep, "unit": "LUP/s", "cores": orc.threads, "cpu_model": cpu_model(), "kind": kind,
           "grid": f"{n}^3",
           "sample": f"{sweeps} sweeps of red_black_gauss_3D at {n}^3 (the benchmarked grid), BCs NDDNDD, "
                     f"OMP_NUM_THREADS={orc.threads} ({usable_cpus(10**6)} usable host CPUs)",
           "ms_per_sweep": per_sweep * 1e3}
    del u, rhs
    # SURVEY 8d also asks for s/V-cycle: ONE pass of the reference's solve loop (V-cycle + update_u) on the
    # timed workload itself (Ax Laplace problem, n^3) - its generic restriction / interpolation dominate
    # it, so this is context for whole-solve ratios, not a smoother comparison
    try:
        t0 = time.perf_counter()
        orc.solve_bvp(u0, np.zeros_like(u0), mesh, "NDDNDD", ms=5, nmax=1)
        out["vcycle_s"] = time.perf_counter() - t0
        out["vcycle_sample"] = f"one V-cycle + update_u of solve_poisson_bvp at {n}^3 (ms=5), Ax Laplace problem, same threads"
    except Exception as exc:  # noqa: BLE001
        out["vcycle_sample"] = f"not timed: {type(exc).__name__}: {exc}"
    return out


def live_traffic(n, zero_rhs, kernel_sub, timeout_s=75):
    """HBM bytes of one launch of the dominant kernel MEASURED IN THIS RUN: two child runs of rocprofv3 (kernel
    trace + ONE counter each - FETCH_SIZE, WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) over
    scripts/run_sweeps.py; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B).
    Returns (bytes per launch, source text) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself being profiled"
    tmp = tempfile.mkdtemp(prefix="ndsm_bench_pmc_")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--",
                   sys.executable, os.path.join(ROOT, "scripts", "run_sweeps.py"), str(n), "7"] + (["zero"] if zero_rhs else [])
            rc_, _o, _e = run_child(cmd, timeout_s, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc_ != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {rc_})"
            got = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                   if row["Counter_Name"] == ctr and kernel_sub in row["Kernel_Name"].replace(" ", "")]
            if not got:
                return None, f"kernel not found in the {ctr} pass"
            vals[ctr] = sum(got) / len(got)
    except Exception as exc:  # noqa: BLE001
        return None, f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0          # the counters are in KiB
    return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child "
                   "passes over scripts/run_sweeps.py), FETCH_SIZE x 2 (gfx950), per-launch average")


def end_to_end_fresh(n):
    """the same two calls in a FRESH process (what a user's first call costs: HIP runtime bring-up and
    code-object load, hierarchy set-up, first touch of the result pages), as a child process"""
    code = ("import sys, json; sys.path.insert(0, %r); import bench, ndsm_amd; L = ndsm_amd.load_library(); "
            "print(json.dumps(bench.end_to_end(L, %d, pretouch=False)))" % (ROOT, n))
    try:
        rc_, out_, err_ = run_child([sys.executable, "-c", code], 600, cwd=ROOT)
        lines = [l for l in out_.splitlines() if l.startswith("{")]
        if rc_ != 0 or not lines:
            return {"error": (err_ or out_)[-300:]}
        j = json.loads(lines[-1])
        return {"first_call_s": j.get("e2e_s"), "second_call_s": j.get("e2e_second_call_s"),
                "what": "fresh process, A = numpy.zeros (untouched pages, as the reference's ndsm.py passes it): the first "
                        "call includes HIP runtime / code-object bring-up and the set-up of the cached hierarchies"}
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def end_to_end(L, n, pretouch=True):
    """ndsm_vector_solve - the reference's actual entry point - at n^3 through raw ctypes, host buffers in
    and out (PCIe inclusive): wall time of the FIRST call in this process and of a SECOND call on the same
    mesh (hierarchy, tables and device pool cached inside the library)."""
    import ctypes
    from golden_inputs import analytic_case
    x, y, z, _A1, b1 = analytic_case(n)
    del _A1
    nshape = np.array([n, n, n, 3], dtype=np.intc)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    L.ndsm_vector_solve.argtypes = [ctypes.c_size_t, ip, ip, dp, dp, dp, dp, dp, dp]
    L.ndsm_vector_solve.restype = ctypes.c_int
    times, ncyc = [], None
    A = np.empty(b1.size) if pretouch else None
    B = np.empty(b1.size)
    for _ in range(2):
        ioptc = np.zeros(16, dtype=np.intc)
        ropt = np.zeros(16)
        ioptc[L.get_iopt_ms()] = 5
        ioptc[L.get_iopt_ncycles()] = 1024
        ioptc[L.get_iopt_iopt_nmaxex()] = 10000
        ioptc[L.get_iopt_dumax()] = 1
        ropt[L.get_ropt_vtol()] = 1e-10
        ropt[L.get_ropt_ctol()] = 1e-13
        if pretouch:
            A[:] = 0.0
        else:
            A = np.zeros(b1.size)       # untouched zero pages, as ndsm.py:176 hands them over
        B[:] = b1.ravel()
        t0 = time.perf_counter()
        ierr = L.ndsm_vector_solve(ctypes.c_size_t(B.size), nshape.ctypes.data_as(ip), ioptc.ctypes.data_as(ip),
                                   ropt.ctypes.data_as(dp), x.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                   z.ctypes.data_as(dp), A.ctypes.data_as(dp), B.ctypes.data_as(dp))
        times.append(time.perf_counter() - t0)
        if ierr != 0:
            return {"error": f"ndsm_vector_solve returned {ierr}"}
        ncyc = int(ioptc[L.get_iopt_ncyc_out()])
    py_s = None
    noisy_s = None
    if pretouch:
        # the analytic field has A_z = 0 (that solve stops after one cycle): the same call on a field whose three
        # components all run their full number of cycles - the analytic one plus noise on the boundary
        rng = np.random.default_rng(3)
        A[:] = 0.0
        B[:] = b1.ravel()
        for c in range(3):
            blk = B[c * n ** 3:(c + 1) * n ** 3]
            blk += 0.2 * rng.standard_normal(blk.size)
        ioptc[:] = 0
        ioptc[L.get_iopt_ms()] = 5
        ioptc[L.get_iopt_ncycles()] = 1024
        ioptc[L.get_iopt_iopt_nmaxex()] = 10000
        ioptc[L.get_iopt_dumax()] = 1
        t0 = time.perf_counter()
        ierr = L.ndsm_vector_solve(ctypes.c_size_t(B.size), nshape.ctypes.data_as(ip), ioptc.ctypes.data_as(ip),
                                   ropt.ctypes.data_as(dp), x.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                   z.ctypes.data_as(dp), A.ctypes.data_as(dp), B.ctypes.data_as(dp))
        noisy_s = {"e2e_s": time.perf_counter() - t0, "ierr": int(ierr), "ncycles_last_3d_solve": int(ioptc[L.get_iopt_ncyc_out()]),
                   "what": "the same call (cached context) on the analytic field + noise: all three 3-D solves iterate"}
        # the reference-compatible Python front end on top (ndsm.py's calling convention: fresh result arrays)
        import ndsm_amd
        del A, B
        t0 = time.perf_counter()
        _ie, _A, _B = ndsm_amd.vector_potential(x, y, z, b1)
        py_s = time.perf_counter() - t0
        del _A, _B
    return {"e2e_s": times[0], "e2e_second_call_s": times[1], "python_front_end_s": py_s, "ncycles_last_3d_solve": ncyc,
            "all_components_iterating": noisy_s,
            "what": f"ndsm_vector_solve at {n}^3, host buffers in and out (the initial guess A up unless it is all zero, 6 GiB of "
                    "A and B down over PCIe, six 2-D + three 3-D solves to vc_tol=1e-10, flux balance, curl); second call = same mesh again"}


def slab_window_problem(n3, sl):
    """window [k0, k0+nloc) ∩ [0, nz) of BASELINE config[3], the Poisson problem SURVEY 8d specifies for it:
    manufactured u* = cos(pi x/Lx) sin(pi y/Ly) sin(pi z/Lz) (x-Neumann, y/z-Dirichlet: BCs NDDNDD), rhs =
    laplace(u*), zero initial guess.  Returns (mesh, rhs window, first global plane of the window)."""
    nx, ny, nz = n3
    x = np.linspace(0.0, 1.0, nx)
    dx = x[1] - x[0]
    y = np.arange(ny) * dx
    z = np.arange(nz) * dx
    a, b = max(sl["k0"], 0), min(sl["k0"] + sl["nloc"], nz)
    kx, ky, kz = np.pi / (x[-1] - x[0]), np.pi / (y[-1] - y[0]), np.pi / (z[-1] - z[0])
    lam = kx * kx + ky * ky + kz * kz
    rhs = (-lam * np.sin(kz * z[a:b]))[:, None, None] * np.sin(ky * y)[None, :, None] * np.cos(kx * x)[None, None, :]
    return [x, y, z], np.ascontiguousarray(rhs), a


def baseline_configs(_lib, L, ms, steps):
    """BASELINE.json configs 0, 1 and 3 on ONE GPU (config[2] is the timed workload itself, config[4] needs the
    8-GPU node): the manufactured Poisson problem SURVEY 8d names for them - time per solve-loop cycle, cycles
    to vc_tol = 1e-10, whole-solve time.  Arrays resident in HBM before each timed region."""
    out = {}
    for key, n3, ngrids, what in (
            ("config0", [64, 64, 64], 3, "64^3 Poisson, 3-level V-cycle (BASELINE config[0]; the reference rule gives 5 levels)"),
            ("config1", [256, 256, 256], 6, "256^3 Poisson fp64, 6-level V-cycle (BASELINE config[1]; the reference rule gives 7)"),
            ("config3_on_one_gpu", [1024, 1024, 512], 0, "1024x1024x512 Poisson fp64 (BASELINE config[3]'s workload) on one GPU, reference level rule")):
        try:
            mesh, rhs, _a = slab_window_problem(n3, {"k0": 0, "nloc": n3[2]})
            S = _lib.MGSolver(n3, mesh, "NDDNDD", ngrids=ngrids, ms=ms, lib=L)
            S.upload(1, _lib.BUF_RHS, rhs)
            del rhs
            S.solve(vc_tol=0.0, nmax=2)
            S.sync()
            t0 = time.perf_counter()
            S.solve(vc_tol=0.0, nmax=steps)
            S.sync()
            cyc_ms = (time.perf_counter() - t0) / steps * 1e3
            ent = {"what": what, "ms_per_cycle": cyc_ms, "levels": S.ngrids,
                   "LUPs_per_s": 2 * ms * float(n3[0]) * n3[1] * n3[2] / (cyc_ms * 1e-3)}
            if key != "config3_on_one_gpu":
                S.upload(1, _lib.BUF_U, np.zeros(tuple(n3[::-1])))
                S.sync()
                t0 = time.perf_counter()
                ie, du, nc, _h = S.solve(vc_tol=1e-10, nmax=1024)
                S.sync()
                ent.update({"cycles_to_tolerance": nc, "solve_s": time.perf_counter() - t0, "du_last": du, "ierr": ie,
                            "vc_tol": 1e-10})
            S.close()
            del S
            out[key] = ent
        except Exception as exc:  # noqa: BLE001
            out[key] = {"what": what, "error": f"{type(exc).__name__}: {exc}"}
    return out


def end_to_end_small(L):
    """ndsm_vector_solve at the sizes the reference's own users run (its integration test: 22^3 ... 220^3,
    tests/integration_test/results_test1.txt:6-14): wall time of the SECOND call on a mesh (cached context), host
    buffers in and out, analytic test field, default options."""
    import ctypes
    from golden_inputs import analytic_case
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    L.ndsm_vector_solve.argtypes = [ctypes.c_size_t, ip, ip, dp, dp, dp, dp, dp, dp]
    L.ndsm_vector_solve.restype = ctypes.c_int
    out = {"what": "ndsm_vector_solve wall time, second call on the same mesh, host buffers in and out, vc_tol=1e-10, ms=5",
           "reference_published_220_s": 174.0,
           "reference_published_source": "tests/integration_test/results_test1.txt:14 (hardware and thread count not stated)"}
    for n in (64, 128, 220):
        try:
            x, y, z, _A1, b1 = analytic_case(n)
            nshape = np.array([n, n, n, 3], dtype=np.intc)
            A, B = np.empty(b1.size), np.empty(b1.size)
            best = None
            for _ in range(3):
                ioptc = np.zeros(16, dtype=np.intc)
                ropt = np.zeros(16)
                ioptc[L.get_iopt_ms()] = 5
                ioptc[L.get_iopt_ncycles()] = 1024
                ioptc[L.get_iopt_iopt_nmaxex()] = 10000
                ioptc[L.get_iopt_dumax()] = 1
                ropt[L.get_ropt_vtol()] = 1e-10
                ropt[L.get_ropt_ctol()] = 1e-13
                A[:] = 0.0
                B[:] = b1.ravel()
                t0 = time.perf_counter()
                ierr = L.ndsm_vector_solve(ctypes.c_size_t(B.size), nshape.ctypes.data_as(ip), ioptc.ctypes.data_as(ip),
                                           ropt.ctypes.data_as(dp), x.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                           z.ctypes.data_as(dp), A.ctypes.data_as(dp), B.ctypes.data_as(dp))
                dt = time.perf_counter() - t0
                if ierr != 0:
                    raise RuntimeError(f"ndsm_vector_solve returned {ierr}")
                if _ > 0:
                    best = dt if best is None else min(best, dt)
            out[f"{n}^3_ms"] = best * 1e3
        except Exception as exc:  # noqa: BLE001
            out[f"{n}^3_ms"] = f"error: {type(exc).__name__}: {exc}"
    return out


def slab_self_check(_lib, L, dist, rank, world):
    """The distributed path against the single-GPU solver on a problem small enough to gather:
    level 1 of a 128 x 128 x 32N box cut into N z-slabs (RCCL halo exchange, the same code path as
    the timed workload, two distributed levels forced), 2 V-cycles, owned planes gathered to rank 0
    over gloo and compared BIT FOR BIT with the same V-cycles of one MGSolver.  Returns a string."""
    import torch
    ns = [128, 128, 32 * world]
    dx = 1.0 / (ns[0] - 1)
    mesh = [np.arange(n) * dx for n in ns]
    rng = np.random.default_rng(2112)
    u = rng.uniform(-1, 1, tuple(ns[::-1]))
    rhs = rng.uniform(-1, 1, tuple(ns[::-1]))
    old = os.environ.get("NDSM_HIP_DIST_LEVELS")
    os.environ["NDSM_HIP_DIST_LEVELS"] = "2"
    try:
        W = _lib.World(ns, mesh, "NDDNDD", world, rank, lib=L)
    finally:
        if old is None:
            os.environ.pop("NDSM_HIP_DIST_LEVELS", None)
        else:
            os.environ["NDSM_HIP_DIST_LEVELS"] = old
    sl = W.slabs[0]
    a, b = max(sl["k0"], 0), min(sl["k0"] + sl["nloc"], ns[2])
    W.upload_window(1, _lib.BUF_U, u[a:b], a)
    W.upload_window(1, _lib.BUF_RHS, rhs[a:b], a)
    W.vcycle(2)
    mine = np.empty((sl["z1"] - sl["z0"], ns[1], ns[0]))
    _lib._check(L.ndsm_hip_world_download(W.h, 1, _lib.BUF_U, mine.ctypes.data_as(_lib._dp)), "download", L)
    levels = W.dist_levels
    W.close()
    parts = [None] * world
    dist.all_gather_object(parts, (sl["z0"], sl["z1"], mine))
    if rank != 0:
        return ""
    got = np.empty_like(u)
    for z0, z1, p in parts:
        got[z0:z1] = p
    S = _lib.MGSolver(ns, mesh, "NDDNDD", lib=L)
    S.upload(1, _lib.BUF_U, u)
    S.upload(1, _lib.BUF_RHS, rhs)
    S.vcycle(2)
    want = S.download(1, _lib.BUF_U)
    S.close()
    nd = int((got != want).sum())
    tag = f"{ns[0]}x{ns[1]}x{ns[2]} in {world} z-slabs over RCCL, {levels} distributed levels, 2 V-cycles"
    if nd == 0:
        return "bit-identical to the single-GPU solver (" + tag + ")"
    return f"MISMATCH: {nd} points differ, max {np.abs(got - want).max():.3e} (" + tag + ")"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="points per dimension of the fine grid (1 GPU)")
    ap.add_argument("--ms", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-solve / ndsm_vector_solve timings")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from profiles/traffic_latest.json instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--slab-shape", default="1024,1024,512",
                    help="N > 1: the grid cut into z-slabs (default: BASELINE config[3]; smaller for rehearsals)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a lone process asked for N GPUs: it becomes the launcher of N ranks (one process per GPU)
        raise SystemExit(spawn_ranks(args))
    if args.gpus > 1 and not os.environ.get("NDSM_BENCH_WORKER"):
        raise SystemExit(supervise_rank())
    attempt = int(os.environ.get("NDSM_BENCH_ATTEMPT", "0"))

    # ONE JSON line on stdout: everything else this process and the native libraries under it write to
    # file descriptor 1 (the reference's and our Fortran `PRINT *` warnings, flushed at exit) goes to
    # stderr; the result line is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # one process per GPU: a rank of an M-rank job must not report an N-GPU number
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...` "
                         f"(or without a launcher: `python3 bench.py --gpus {args.gpus}` starts its own ranks)")

    # libndsm_hip FIRST: it loads /opt/rocm's libamdhip64.so.7 / librccl.so.1.  PyTorch bundles a
    # second ROCm stack under the same SONAMEs; whichever is loaded first serves the whole process,
    # so torch is imported afterwards and used for its CPU (gloo) rendezvous only - the device is
    # synchronised through the library's own stream (ndsm_hip_sync), not torch.cuda.synchronize().
    import ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library()
    ndev = L.ndsm_hip_device_count()
    if world > 1 and ndev < world and not os.environ.get("NDSM_HIP_LIB"):   # (NDSM_HIP_LIB: the test double's rehearsal)
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible - one process per GPU")
    rc = L.ndsm_hip_init(local_rank % max(1, ndev))
    if rc != 0:
        raise SystemExit("libndsm_hip: " + _lib.last_error(L))

    dist = None
    slab_check = None
    rccl_ranks = 0
    if world > 1:
        import faulthandler
        faulthandler.dump_traceback_later(600, exit=True)   # a wedged collective must not hang the node
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node by contract: the CPU rendezvous / control traffic stays on the loop-back interface
            # (gloo otherwise picks the interface the host NAME resolves to - which it may not)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if attempt == 0:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            # second worker of this rank: a store of its own (rank 0 hosts it), not the launcher's, which still holds
            # the first worker's keys
            port2 = 20000 + (int(os.environ.get("MASTER_PORT", "29500")) + 7919) % 20000
            dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{port2}", rank=rank, world_size=world)
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(_lib.dist_unique_id(L)), dtype=torch.uint8).clone()
        dist.broadcast(uid, 0)

    def all_ok(ok, what, err=""):
        """every rank must succeed: the z-slab path either runs on all N GPUs or the bench FAILS - it never
        degrades to N independent replicas behind an n_gpus = N line"""
        import torch
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0]) < 0.5:
            print(f"bench.py[rank {rank}]: {what} failed" + (f": {err}" if err else " on another rank") +
                  " - no result line (the z-slab RCCL path did not run)", file=sys.stderr, flush=True)
            dist.barrier()
            raise SystemExit(3)

    def barrier_sync():
        _lib._check(L.ndsm_hip_sync(), "sync", L)
        if dist is not None:
            dist.barrier()

    ms = args.ms
    timed_dog = None
    if world == 1:
        n = args.n
        n3 = [n, n, n]
        mesh, u0 = boundary_problem(n)
        S = _lib.MGSolver(n3, mesh, "NDDNDD", ms=ms)
        S.upload(1, _lib.BUF_U, u0)
        S.zero_rhs()       # the vector potential's 3-D problems are Laplace problems (rhs = 0, :640-641)
        # one step = one pass of the reference's solve loop (solve_poisson_bvp, ndsm_poisson.f90:104-150):
        # V-cycle + update_u's max|u_new - u_old| + the host's strict du < vc_tol test; vc_tol = 0 is never
        # met, so exactly K cycles run
        run_cycles = lambda k: S.solve(vc_tol=0.0, nmax=k)  # noqa: E731
        ngrids = S.ngrids
        workload = (f"{n}^3 vector-potential Ax component, one V-cycle + convergence metric per step (ms={ms}, "
                    f"{ngrids} grids), config[2] of BASELINE.json")
        parallelism = "single GPU"
        scaling = "weak"
        slab_mode = False
    else:
        # BASELINE config[3]: 1024 x 1024 x 512 Poisson, level 1 z-slab decomposed over the ranks,
        # RCCL halo exchange between sweeps, coarse levels distributed while a rank's share is large,
        # the rest on rank 0
        err = ""
        try:
            _lib.dist_init(rank, world, uid.numpy().tobytes(), L)     # RCCL communicator over xGMI
            rccl_ranks = _lib.dist_info(L)[1]
            if rccl_ranks != world:
                raise RuntimeError(f"RCCL communicator reports {rccl_ranks} ranks, expected {world}")
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        all_ok(not err, "RCCL communicator bring-up", err)
        # the distributed path against the single-GPU solver on the real transport, bit for bit - with the
        # halo exchange on the main stream and (forced) on the second stream behind the interior planes
        checks = []
        hang_s = float(os.environ.get("NDSM_BENCH_OVERLAP_TIMEOUT", "180"))
        for ov in ("0", "1"):
            if ov == "1" and attempt > 0:
                checks.append("MISMATCH: the overlapped exchange did not come back within "
                              f"{hang_s:.0f} s in this job's first worker; not tried again")
                continue
            # from here until the timed region is over a wedged overlapped exchange ends this worker with
            # EXIT_OVERLAP_HUNG (supervise_rank starts the next one without the overlap)
            if ov == "1":
                overlap_dog = Watchdog(hang_s, EXIT_OVERLAP_HUNG, "the overlapped self-check")
            old = os.environ.get("NDSM_HIP_OVERLAP")
            os.environ["NDSM_HIP_OVERLAP"] = ov
            try:
                if ov == "1" and os.environ.get("NDSM_BENCH_FAKE_HANG"):   # (tests/test_gpu_multirank.py)
                    time.sleep(1e6)
                checks.append(slab_self_check(_lib, L, dist, rank, world))
            except Exception as exc:  # noqa: BLE001
                checks.append(f"MISMATCH: self-check could not run: {type(exc).__name__}: {exc}")
            finally:
                if old is None:
                    os.environ.pop("NDSM_HIP_OVERLAP", None)
                else:
                    os.environ["NDSM_HIP_OVERLAP"] = old
        # verdicts are rank 0's (it holds the gathered planes): every rank must act on the same ones
        verdict = [checks if rank == 0 else None]
        dist.broadcast_object_list(verdict, 0)
        checks = verdict[0]
        if attempt == 0:
            overlap_dog.cancel()
        all_ok("MISMATCH" not in checks[0], "slab self-check against the single-GPU solver (exchange on the main stream)", checks[0])
        slab_check = checks[0] + " [exchange on the main stream]; " + checks[1] + " [exchange overlapped]"
        if "MISMATCH" in checks[1]:
            # the overlapped schedule (second stream, one communicator) misbehaves on this node but the
            # single-stream one is right: time THAT - still the z-slab RCCL path - and say so
            os.environ["NDSM_HIP_OVERLAP"] = "0"
            slab_check += "; TIMED WITH NDSM_HIP_OVERLAP=0 (single-stream exchange) because the overlapped self-check failed"
        if attempt == 0 and os.environ.get("NDSM_HIP_OVERLAP", "1") != "0":
            timed_dog = Watchdog(300.0, EXIT_OVERLAP_HUNG, "the z-slab world's set-up and timed region with the overlapped exchange")
        n3 = [int(v) for v in args.slab_shape.split(",")]
        x = np.linspace(0.0, 1.0, n3[0])
        dx = x[1] - x[0]
        mesh = [x, np.arange(n3[1]) * dx, np.arange(n3[2]) * dx]
        S, err = None, ""
        try:
            S = _lib.World(n3, mesh, "NDDNDD", world, rank, ms=ms, lib=L)
            _m, win, a = slab_window_problem(n3, S.slabs[0])
            S.upload_window(1, _lib.BUF_RHS, win, a)              # general right-hand side in HBM: 24 B/LUP
            del win                                               # (u starts at zero: the initial guess and, sin
                                                                  # vanishing there, the Dirichlet data)
            S.vcycle(1)
            S.sync()
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        all_ok(not err, f"z-slab world ({n3[0]}x{n3[1]}x{n3[2]})", err)
        run_cycles = lambda k: S.solve(vc_tol=0.0, nmax=k)  # noqa: E731
        # Which schedule is faster on THIS node - the halo exchange overlapped with the interior planes (every pass in
        # three launches) or on the main stream (one launch per pass) - depends on what the links deliver; DESIGN
        # section 6's model puts them within a few per cent of each other.  Both passed the self-check: a short trial
        # outside the timed region decides (the library reads NDSM_HIP_OVERLAP at every pass; 0 / 1 force it).
        if "MISMATCH" not in checks[1] and os.environ.get("NDSM_HIP_OVERLAP") is None:
            import torch
            trial = {}
            for ov in ("0", "1"):
                os.environ["NDSM_HIP_OVERLAP"] = ov
                run_cycles(1)
                barrier_sync()
                t0 = time.perf_counter()
                run_cycles(3)
                barrier_sync()
                t = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                trial[ov] = float(t[0]) * 1e3
            best = min(trial, key=trial.get)       # (the all-reduced times: every rank picks the same one)
            os.environ["NDSM_HIP_OVERLAP"] = best
            slab_check += (f"; schedule for the timed region chosen by a 3-cycle trial: NDSM_HIP_OVERLAP={best} "
                           f"({trial['0']:.2f} ms per cycle on the main stream, {trial['1']:.2f} overlapped)")
        ngrids = 8
        workload = (f"{n3[0]}x{n3[1]}x{n3[2]} Poisson (manufactured right-hand side, zero initial guess: SURVEY 8d), one "
                    f"V-cycle + convergence metric per step (ms={ms})" +
                    (", config[3] of BASELINE.json" if n3 == [1024, 1024, 512] else ", REHEARSAL SHAPE (not config[3])"))
        parallelism = (f"level 1 in {world} z-slabs (RCCL send/recv halo: 4 planes per neighbour per two-sweep "
                       f"pass), {S.dist_levels} distributed level(s), the rest on rank 0")
        scaling = "strong"
        slab_mode = True
    npts = float(n3[0]) * n3[1] * n3[2]

    # ---- the timed region: K V-cycles ------------------------------------
    run_cycles(args.warmup)
    barrier_sync()
    t0 = time.perf_counter()
    run_cycles(args.steps)
    barrier_sync()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])
    if timed_dog is not None:
        timed_dog.cancel()
    ms_per_step = el / args.steps * 1e3

    # ---- dominant kernel: the level-1 smoother launches under HIP events on the library stream ----
    nsw = 20
    vc_only_ms = gen_ms = rs_ms = sweeps_per_cycle = one_gpu_ms = None
    solve_info = None
    if world == 1:
        S.vcycle(2)
        S.sync()
        vc_only_ms = S.timed(lambda: S.vcycle(args.steps)) / args.steps   # the cycle without update_u
        # the kernel the timed V-cycles spend most of their time in: the two-sweep LAPLACE launch (rhs
        # declared zero and never read: 16 B/LUP algorithmic) ...
        S.op(_lib.OP_RELAX, 1, 2)
        S.sync()
        lap_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw
        sweeps, _unconv = S.info()
        sweeps_per_cycle = sweeps / max(1, args.steps * 2 + args.warmup + 2)
        if not args.no_e2e:
            # whole Ax solve to the reference's default tolerance (SURVEY 8d "total solve time")
            S.upload(1, _lib.BUF_U, u0)
            S.sync()
            t0 = time.perf_counter()
            ie, du, nc, _h = S.solve(vc_tol=1e-10, nmax=1024)
            S.sync()
            solve_info = {"solve_s": time.perf_counter() - t0, "ncycles": nc, "du_last": du, "ierr": ie,
                          "what": f"{n}^3 Ax Laplace solve to vc_tol=1e-10 on the resident hierarchy (device time + 16-byte read-backs)"}
        # ... and the general kernel with a right-hand side in HBM (level 2 of these solves; level 1 of
        # Poisson problems): 24 B/LUP
        S.upload(1, _lib.BUF_RHS, np.random.default_rng(2113).uniform(-1, 1, (n, n, n)))
        S.op(_lib.OP_RELAX, 1, 2)
        S.sync()
        gen_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw
        rs_ms = S.timed(lambda: [S.op(_lib.OP_RESIDUAL, 1) for _ in range(5)]) / 5
        bpl = 16.0
        kname = "rbgs3_fused_k<double, 2, 136, 30, 1024, 4, true, 0, true, false>"
    else:
        S.relax(2)
        barrier_sync()
        lap_ms = S.timed(lambda: S.relax(nsw)) / nsw               # includes the halo exchanges
        import torch
        t = torch.tensor([lap_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lap_ms = float(t[0])
        bpl = BYTES_PER_LUP
        kname = "rbgs3_fused_k<double, 2, 136, 30, 1024, 4, false, 0, *> on z-slab windows, halo exchange included"
        # the SAME workload on one GPU (rank 0's, the others wait): the driver's N = 1 line is the 512^3
        # Laplace problem, so the strong-scaling ratio of this configuration needs its own one-GPU time
        if rank == 0 and not args.no_e2e:
            try:
                S1 = _lib.MGSolver(n3, mesh, "NDDNDD", ms=ms, lib=L)
                _m, full, _a = slab_window_problem(n3, {"k0": 0, "nloc": n3[2]})
                S1.upload(1, _lib.BUF_RHS, full)
                del full
                S1.solve(vc_tol=0.0, nmax=2)
                S1.sync()
                t0 = time.perf_counter()
                S1.solve(vc_tol=0.0, nmax=args.steps)
                S1.sync()
                one_gpu_ms = (time.perf_counter() - t0) / args.steps * 1e3
                S1.close()
                del S1
            except Exception as exc:  # noqa: BLE001
                one_gpu_ms = None
                print(f"bench: one-GPU time of the slab workload not taken: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
    S.close()
    del S

    # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
    # MI355X_MICROARCH.md) - NOT measured in this run
    traffic = traffic_gen = None
    tsrc = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world == 1 and args.n == 512:
        try:
            tj = json.load(open(tpath))
            for k, v in tj.get("kernels", {}).items():
                if k.startswith("zero <double, 2"):
                    traffic = v["total_bytes"]
                if k.startswith("general <double, 2"):
                    traffic_gen = v["total_bytes"]
            tsrc = "profiles/traffic_latest.json (" + tj.get("collected", "rocprofv3 --pmc, separate passes") + "); not measured in this run"
        except Exception:  # noqa: BLE001
            traffic = traffic_gen = None

    # ... unless the counters can be read right now (rank 0, one GPU, the 512^3 workload): then `traffic` IS measured
    # in this run, by two short child runs under rocprofv3
    tsrc_gen = tsrc
    if world == 1 and args.n == 512 and not args.no_live_traffic:
        t_live, src = live_traffic(args.n, True, "rbgs3_fused_k<double,2,136,30,1024,4,true,0,true,false>")
        if t_live:
            traffic, tsrc = t_live, src
            g_live, gsrc = live_traffic(args.n, False, "rbgs3_fused_k<double,2,136,30,1024,4,false,0,true,false>")
            if g_live:
                traffic_gen, tsrc_gen = g_live, gsrc
        else:
            tsrc = (tsrc or "") + f" [live measurement skipped: {src}]"

    def roofline(ms_sweep, bytes_per_lup, kernel, tr, src=None):
        launch_s = 2 * ms_sweep * 1e-3
        alg = 2 * bytes_per_lup * npts / world                    # per launch and GPU: two sweeps
        # achieved / frac: what the memory system really moved (HBM bytes per launch from the rocprofv3 counters)
        # over the launch time; null when no counter figure exists for this launch.  The algorithmic figure
        # stands beside it under its own name: ONE launch performs TWO sweeps on one pass over HBM (temporal
        # blocking), so bytes_per_lup x 2 sweeps x points / time says how fast the sweeps go, not what HBM did -
        # SURVEY 8d's 24 B/LUP (u in, rhs in, u out per sweep) does not describe a launch that reads u once for two
        # sweeps and (Laplace variant) never reads rhs.
        hbm = (tr / launch_s / 1e9) if tr else None
        r = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (hbm / HBM_PEAK_GBS) if hbm else None, "traffic": tr,
             "kernel": kernel, "avg_launch_ms": launch_s * 1e3,
             "traffic_source": (src if src is not None else tsrc) if tr else None,
             "bytes_per_lup": bytes_per_lup, "sweeps_per_launch": 2, "algorithmic_bytes_per_launch": alg,
             "achieved_algorithmic": alg / launch_s / 1e9,
             "frac_algorithmic": alg / launch_s / 1e9 / HBM_PEAK_GBS,
             "note": "frac = achieved / peak with achieved = HBM bytes the launch really moved (traffic: FETCH_SIZE x 2 + "
                     "WRITE_SIZE, rocprofv3 --pmc) / average launch time (HIP events on the library stream); "
                     "frac_algorithmic = bytes_per_lup x 2 sweeps x points / launch time / peak - it exceeds frac because "
                     "one pass over HBM carries two sweeps" +
                     ("" if tr else "; NO counter figure for this launch: frac is null, only the algorithmic figure is given")}
        return r

    def finish():
        # after the result line is out: leave together, then take the communicator down (collective)
        if dist is not None:
            dist.barrier()
            try:
                _lib.dist_finalize(L)
            except Exception as exc:  # noqa: BLE001
                print(f"bench: dist_finalize: {exc}", file=sys.stderr, flush=True)

    if rank != 0:
        finish()
        return
    out = {
        "metric": "fine-grid LUP/s (RB-GS smoother updates per second of V-cycle time, fp64)",
        "value": 2 * ms * npts / (ms_per_step * 1e-3),
        "unit": "LUP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "global_points": int(npts), "parallelism": parallelism},
        "slab_mode": slab_mode,
        "rccl_ranks": rccl_ranks,
        "vcycles_per_s": 1.0 / (ms_per_step * 1e-3),
        "vcycle_without_metric_ms": vc_only_ms,
        "smoother": ({"laplace_ms_per_sweep": lap_ms, "laplace_LUPs_per_s": npts / (lap_ms * 1e-3),
                      "general_rhs_ms_per_sweep": gen_ms,
                      "general_rhs_LUPs_per_s": (npts / (gen_ms * 1e-3)) if gen_ms else None,
                      "residual_ms": rs_ms} if world == 1 else
                     {"general_rhs_ms_per_sweep": lap_ms, "general_rhs_LUPs_per_s": npts / (lap_ms * 1e-3),
                      "note": "whole job, halo exchanges included"}),
        "coarse_exact_sweeps_per_cycle": sweeps_per_cycle,
        # the kernel that dominates the timed region (top row of profiles/*_kernel_stats.csv)
        "roofline": roofline(lap_ms, bpl, kname, traffic),
        "rocm_stack": _lib.bound_libs(L),
    }
    if gen_ms:
        out["roofline_general_rhs"] = roofline(gen_ms, BYTES_PER_LUP,
                                               "rbgs3_fused_k<double, 2, 136, 30, 1024, 4, false, 0, true, false>", traffic_gen,
                                               tsrc_gen)
    if solve_info:
        out["solve"] = solve_info
    if slab_check is not None:
        out["slab_check"] = slab_check
    if one_gpu_ms:
        out["same_workload_on_one_gpu"] = {"ms_per_step": one_gpu_ms, "speedup": one_gpu_ms / ms_per_step,
                                           "what": f"the identical {n3[0]}x{n3[1]}x{n3[2]} problem and loop on rank 0's GPU alone, "
                                                   "timed after the N-GPU region (strong-scaling reference)"}
    if world == 1 and not args.no_e2e:
        out["configs"] = baseline_configs(_lib, L, ms, max(3, min(args.steps, 10)))
        try:
            out["end_to_end_small"] = end_to_end_small(L)
        except Exception as exc:  # noqa: BLE001
            out["end_to_end_small"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            out["end_to_end"] = end_to_end(L, args.n)
            out["end_to_end"]["fresh_process"] = end_to_end_fresh(args.n)
        except Exception as exc:  # noqa: BLE001
            out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.n)
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    finish()


if __name__ == "__main__":
    main()
