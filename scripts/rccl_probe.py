"""Can two ranks share ONE GPU under RCCL here?  If yes the RCCL transport of the z-slab world can be
exercised on the 1-GPU box (correctness only).  usage: torchrun-style env, 2 procs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    uid = torch.frombuffer(bytearray(_lib.dist_unique_id()), dtype=torch.uint8).clone()
dist.broadcast(uid, 0)
_lib.dist_init(rank, world, bytes(uid.numpy().tobytes()))
print(rank, "rccl comm up", flush=True)
ns = [64, 64, 64]; mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
W = _lib.World(ns, mesh, "NDDNDD", world, rank)
W.upload(_lib.BUF_U, u); W.upload(_lib.BUF_RHS, rhs)
W.vcycle(2); W.sync()
mine = W.download(_lib.BUF_U)
sl = W.slabs[0]
S = _lib.MGSolver(ns, mesh, "NDDNDD"); S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs); S.vcycle(2)
ref = S.download(1, _lib.BUF_U)
ok = np.array_equal(mine[sl["z0"]:sl["z1"]], ref[sl["z0"]:sl["z1"]])
ierr, du, nc, hist = W.solve(hist_len=32)
print(rank, "slab", sl["z0"], sl["z1"], "vcycle bitwise:", ok, "solve:", ierr, nc, du, flush=True)
dist.barrier()
