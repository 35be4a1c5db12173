"""Seeded input generators shared by the tests and tests/golden/make_golden.py.

The golden files hold only the reference's OUTPUTS; the matching inputs are
re-created here from fixed seeds (numpy default_rng,
seeds 2112 / 2113, U(-1,1)).
"""
import numpy as np

# BC letter sets of the three vector-potential components, lower x,y,z then
# upper x,y,z (ndsm_vector_potential.f90:655,671,687)
BCS3 = ("NDDNDD", "DNDDND", "DDNDDN")

KERNEL_SHAPES_3D = ([22, 22, 22], [33, 22, 27])   # Fortran order [nx, ny, nz]
KERNEL_SHAPES_2D = ([27, 36], [22, 22])


def rand_field(shape, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, size=shape)


def uniform_mesh(nshape):
    """x = linspace(0,1,nx); y,z = arange(n)*dx  (tests/integration_test/integration_test1.py:124-127)."""
    x = np.linspace(0.0, 1.0, int(nshape[0]))
    dx = x[1] - x[0]
    return [x] + [np.arange(int(n)) * dx for n in nshape[1:]]


def manufactured_poisson(mesh, bcs):
    """u* = prod_d (cos|sin)(pi q_d / L_d): cos on the Neumann axis, sin on
    Dirichlet axes; rhs = laplace(u*).  Returned in numpy order (nz, ny, nx)."""
    nd = len(mesh)
    grids = np.meshgrid(*mesh[::-1], indexing="ij")[::-1]  # X, Y, Z each (nz,ny,nx)
    u = np.ones_like(grids[0])
    lam = 0.0
    for d in range(nd):
        L = mesh[d][-1] - mesh[d][0]
        k = np.pi / L
        if bcs[d] == "N":
            assert bcs[nd + d] == "N"
            u = u * np.cos(k * (grids[d] - mesh[d][0]))
        else:
            u = u * np.sin(k * (grids[d] - mesh[d][0]))
        lam += k * k
    return u, -lam * u


def analytic_case(n):
    """Current-free test field of tests/integration_test/integration_test1.py:57-99
    (k = pi, l = sqrt(2) pi) on x = linspace(0,1,nx), equal spacing in y, z.
    `n` is an int (cube) or Fortran-order [nx, ny, nz].  Returns x, y, z, A, b
    with A, b shaped (3, nz, ny, nx)."""
    ns = [n, n, n] if np.isscalar(n) else list(n)
    x, y, z = uniform_mesh(ns)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wn = np.pi
    l = np.sqrt(2 * wn ** 2)
    b = np.zeros((3,) + X.shape)
    A = np.zeros((3,) + X.shape)
    b[0] = +l * np.sin(wn * X) * np.cos(wn * Y) * np.exp(-l * Z)
    b[1] = +l * np.cos(wn * X) * np.sin(wn * Y) * np.exp(-l * Z)
    b[2] = +2 * wn * np.cos(wn * X) * np.cos(wn * Y) * np.exp(-l * Z)
    A[0] = -np.cos(wn * X) * np.sin(wn * Y) * np.exp(-l * Z)
    A[1] = +np.sin(wn * X) * np.cos(wn * Y) * np.exp(-l * Z)
    return x, y, z, A, b
