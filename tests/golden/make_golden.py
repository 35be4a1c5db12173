#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every expected output below is produced by the reference's own Fortran,
compiled unmodified into oracle/_ref/ (oracle/Makefile), either through its
public C ABI (`ndsm_vector_solve`, ndsm_python_wrapper.f90:56) or through
oracle/ref_shim.f90, a pass-through to its PUBLIC module procedures.  Inputs
are re-creatable from the seeds stored next to the outputs (see
tests/golden_inputs.py), so the files stay small.  Nothing here is reference
source text - only numbers.

OMP_NUM_THREADS is forced to 1 so the all-Neumann `mean` reduction
(ndsm_multigrid_core.f90:1214) is summed in serial order.
"""
import json
import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from golden_inputs import (BCS3, analytic_case, manufactured_poisson, rand_field,  # noqa: E402
                           KERNEL_SHAPES_3D, KERNEL_SHAPES_2D)
from oracle import Oracle, have_ref, uniform_mesh  # noqa: E402


def main():
    assert have_ref(), "build the reference first: make -C oracle ref"
    R = Oracle("ref")

    # ---- per-operator vectors, 3-D --------------------------------------
    for ns in KERNEL_SHAPES_3D:
        tag = "x".join(str(n) for n in ns)
        mesh = uniform_mesh(ns)
        shp = tuple(ns[::-1])
        u = rand_field(shp, 2112)
        rhs = rand_field(shp, 2113)
        out = {}
        shapes, meshes = R.hierarchy(ns, mesh)
        out["level_shapes"] = shapes
        for l, lv in enumerate(meshes):
            for d, m in enumerate(lv):
                out[f"mesh_l{l+1}_d{d+1}"] = m
        for bcs in BCS3:
            out[f"relax_{bcs}"] = R.relax3d(u, rhs, mesh, bcs)
            out[f"residual_{bcs}"] = R.residual3d(u, rhs, mesh, bcs)
            out[f"vcycle_{bcs}"] = R.vcycle(u, rhs, mesh, bcs)
        for lvl in range(1, len(shapes)):
            f = rand_field(tuple(int(v) for v in shapes[lvl - 1][::-1]), 3000 + lvl)
            c = rand_field(tuple(int(v) for v in shapes[lvl][::-1]), 4000 + lvl)
            out[f"restrict_l{lvl}"] = R.restrict(f, ns, mesh, lvl)
            out[f"interp_l{lvl}"] = R.interp(c, ns, mesh, lvl)
        un = u.copy()
        out["update_u"] = np.array(R.update_u(rhs, un))
        np.savez(os.path.join(HERE, f"kernels3d_{tag}.npz"), **out)
        print("wrote kernels3d_" + tag)

    # ---- per-operator vectors, 2-D (generic N-D path) --------------------
    for ns in KERNEL_SHAPES_2D:
        tag = "x".join(str(n) for n in ns)
        mesh = uniform_mesh(ns)
        shp = tuple(ns[::-1])
        u = rand_field(shp, 2112)
        rhs = rand_field(shp, 2113)
        rhs0 = rhs - rhs.mean()
        out = {}
        shapes, _ = R.hierarchy(ns, mesh)
        out["level_shapes"] = shapes
        for bcs in ("NNNN", "DNND"):
            out[f"relax_{bcs}"] = R.relax_nd(u, rhs, mesh, bcs)
            out[f"residual_{bcs}"] = R.residual_nd(u, rhs, mesh, bcs)
        out["vcycle_NNNN"] = R.vcycle(u, rhs0, mesh, "NNNN")
        ierr, us, du = R.solve_bvp(np.zeros(shp), rhs0, mesh, "NNNN")
        out["solve_NNNN"] = us
        out["solve_NNNN_meta"] = np.array([ierr, du])
        for lvl in range(1, len(shapes)):
            f = rand_field(tuple(int(v) for v in shapes[lvl - 1][::-1]), 3000 + lvl)
            c = rand_field(tuple(int(v) for v in shapes[lvl][::-1]), 4000 + lvl)
            out[f"restrict_l{lvl}"] = R.restrict(f, ns, mesh, lvl)
            out[f"interp_l{lvl}"] = R.interp(c, ns, mesh, lvl)
        np.savez(os.path.join(HERE, f"kernels2d_{tag}.npz"), **out)
        print("wrote kernels2d_" + tag)

    # ---- full Poisson solves (manufactured solution), history of du ------
    hist = {}
    for ns in ([22, 22, 22], [33, 22, 27], [64, 64, 64]):
        tag = "x".join(str(n) for n in ns)
        mesh = uniform_mesh(ns)
        for bcs in BCS3:
            us, rhs = manufactured_poisson(mesh, bcs)
            u = np.zeros_like(us)
            dus = []
            # the reference only prints du; replay its loop (ndsm_poisson.f90:116-141)
            # one v_cycle + update_u at a time to record it
            prev = u.copy()
            for it in range(64):
                cur = R.vcycle(prev, rhs, mesh, bcs)
                d = float(np.abs(cur - prev).max())
                dus.append(d)
                prev = cur
                if d < 1e-10:
                    break
            ierr, uref, du_last = R.solve_bvp(u, rhs, mesh, bcs)
            assert ierr == 0 and np.array_equal(uref, prev) and du_last == dus[-1]
            key = f"{tag}_{bcs}"
            hist[key] = {"du": dus, "ncycles": len(dus), "err_vs_exact": float(np.abs(uref - us).max())}
            if ns[0] <= 33:
                np.save(os.path.join(HERE, f"solve3d_{key}.npy"), uref)
            else:  # 64^3: keep three orthogonal mid-planes only
                np.savez(os.path.join(HERE, f"solve3d_{key}_planes.npz"), kz=uref[ns[2] // 2],
                         jy=uref[:, ns[1] // 2], ix=uref[:, :, ns[0] // 2])
    with open(os.path.join(HERE, "solve3d_history.json"), "w") as fh:
        json.dump(hist, fh, indent=1)
    print("wrote solve3d_*")

    # ---- full pipeline through the reference C ABI -----------------------
    rows = {}
    for n in (22, 44):
        x, y, z, A1, b1 = analytic_case(n)
        ierr, A, B, ioptc, ropt = R.vector_potential(x, y, z, b1)
        eA = np.linalg.norm(A1 - A, axis=0)
        eB = np.linalg.norm(b1 - B, axis=0)
        rows[str(n)] = {"ierr": int(ierr), "dx": float(x[1] - x[0]), "Ea_max": float(eA.max()),
                        "Ea_avg": float(eA.mean()), "Eb_max": float(eB.max()), "Eb_avg": float(eB.mean())}
        if n == 22:
            np.savez(os.path.join(HERE, "pipeline_22.npz"), A=A, B=B, ioptc=ioptc)
    # anisotropic shape, equal spacing (quirk Q4 needs dx=dy=dz)
    ns = [33, 22, 27]
    x, y, z, A1, b1 = analytic_case(ns)
    ierr, A, B, ioptc, ropt = R.vector_potential(x, y, z, b1)
    np.savez(os.path.join(HERE, "pipeline_33x22x27.npz"), A=A, B=B, ioptc=ioptc)
    rows["33x22x27"] = {"ierr": int(ierr)}
    with open(os.path.join(HERE, "pipeline_rows.json"), "w") as fh:
        json.dump(rows, fh, indent=1)
    print("wrote pipeline_*")


if __name__ == "__main__":
    main()
