"""Worker of tests/test_slab_cpu.py: one rank of a z-slab decomposed smoother on the CPU.

The compute is a numpy statement of the red-black sweep restricted to a slab
(test code, fp64, reference operand order); the DECOMPOSITION - who owns which
planes, how deep the ghosts are, one 2-plane exchange per full sweep with the
red update of the first ghost plane recomputed locally, global colouring through
k0, mirror faces only at the physical boundary - is the product's: the plan comes
from libndsm_hip (ndsm_hip_slab_plan, pure host code) and the exchange pattern is
ndsmh_world.f90's, carried here by torch.distributed (gloo) send/recv.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ndsm_amd  # noqa: E402
from golden_inputs import rand_field, uniform_mesh  # noqa: E402


def color_pass(u, rhs, w, w1, lb, ub, par, k0, nzg, planes):
    """update points of colour `par` ((i+j+kglobal)&1 == par) on local planes `planes`"""
    nz, ny, nx = u.shape
    i = np.arange(nx)[None, :]
    j = np.arange(ny)[:, None]
    xl = np.where(i - 1 < 0, 1, i - 1) + 0 * j
    xh = np.where(i + 1 > nx - 1, nx - 2, i + 1) + 0 * j
    yl = np.where(j - 1 < 0, 1, j - 1) + 0 * i
    yh = np.where(j + 1 > ny - 1, ny - 2, j + 1) + 0 * i
    jj = j + 0 * i
    ii = i + 0 * j
    inb = (ii >= lb[0]) & (ii <= ub[0]) & (jj >= lb[1]) & (jj <= ub[1])
    for k in planes:
        kg = k + k0
        if kg < lb[2] or kg > ub[2]:
            continue
        zl = k + 1 if kg - 1 < 0 else k - 1
        zh = k - 1 if kg + 1 > nzg - 1 else k + 1
        p = u[k]
        new = w1 * ((p[jj, xh] + p[jj, xl]) * w[0] + (p[yh, ii] + p[yl, ii]) * w[1]
                    + (u[zh] + u[zl]) * w[2] - rhs[k])
        m = inb & (((ii + jj + kg) & 1) == par)
        p[m] = new[m]


def main():
    ns = [int(v) for v in sys.argv[1].split("x")]
    bcs = sys.argv[2]
    nsweeps = int(sys.argv[3])
    outdir = sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mesh = uniform_mesh(ns)
    shp = tuple(ns[::-1])
    ug, rg = rand_field(shp, 2112), rand_field(shp, 2113)

    plan = ndsm_amd.slab_plan(ns, mesh, world)[rank]          # the product's plan (host code)
    z0, z1, g, k0, nloc = plan["z0"], plan["z1"], plan["g"], plan["k0"], plan["nloc"]
    nown = z1 - z0
    u = np.zeros((nloc,) + shp[1:])
    rhs = np.zeros_like(u)
    a, b = max(k0, 0), min(k0 + nloc, ns[2])
    rhs[a - k0:b - k0] = rg[a:b]
    u[g:g + nown] = ug[z0:z1]                                  # ghosts arrive by exchange only

    h = [mesh[d][1] - mesh[d][0] for d in range(3)]
    w = [1.0 / (h[d] * h[d]) for d in range(3)]
    w1 = 2 * (w[0] + w[1] + w[2])
    w1 = 1.0 / w1
    lb = [1 if bcs[d] == "D" else 0 for d in range(3)]
    ub = [ns[d] - 1 - (1 if bcs[3 + d] == "D" else 0) for d in range(3)]
    first = 1 if bcs[0] == "D" else 0

    def exchange(depth=2):
        reqs = []
        if rank + 1 < world:
            reqs.append(dist.isend(torch.from_numpy(u[g + nown - depth:g + nown].copy()), rank + 1))
        if rank > 0:
            reqs.append(dist.isend(torch.from_numpy(u[g:g + depth].copy()), rank - 1))
        if rank + 1 < world:
            t = torch.empty((depth,) + shp[1:], dtype=torch.float64)
            dist.recv(t, rank + 1)
            u[g + nown:g + nown + depth] = t.numpy()
        if rank > 0:
            t = torch.empty((depth,) + shp[1:], dtype=torch.float64)
            dist.recv(t, rank - 1)
            u[g - depth:g] = t.numpy()
        for r in reqs:
            r.wait()

    for _ in range(nsweeps):
        exchange(2)
        # fused-sweep schedule: red on the owned planes and one ghost plane each side
        # (it needs the second ghost plane), black on the owned planes only
        color_pass(u, rhs, w, w1, lb, ub, first, k0, ns[2], range(max(g - 1, -k0), min(g + nown + 1, ns[2] - k0)))
        color_pass(u, rhs, w, w1, lb, ub, first ^ 1, k0, ns[2], range(g, g + nown))
    np.save(os.path.join(outdir, f"slab_{rank}.npy"), u[g:g + nown])
    np.save(os.path.join(outdir, f"plan_{rank}.npy"), np.array([z0, z1, g]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
