#!/usr/bin/env python3
"""CPU prototype (test infrastructure: oracle + scipy; nothing of the product): PCG iterations to the reference's tolerance of
  z = M_tile^-1 r + P0 C(P0^T r)
for several coarse spaces P0 and coarse solvers C, on a system assembled by the oracle.
usage: mg_proto.py SIZE [half_tank|dam_break] [preroll frames] [variants ...]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from oracle_lib import C  # noqa: E402
from euler_amd import scenarios  # noqa: E402


def state_before_projection(N, wl, pre):
    o = oracle_lib.Oracle(N, N, fast=True)
    if wl == "half_tank":
        o.load_half_tank()
    else:
        o.load_text(getattr(scenarios, wl)(), upscale=True)
    o.c.tile_records = 16
    o.c.coarse_m = o.lib.eo_coarse_m(N, N)
    o.c.coarse_mg = 1
    o.c.max_iterations = 100
    for _ in range(pre):
        o.step()
    dt = o.timestep(0.1)
    L = o.lib
    f = C.c_float(dt)
    L.eo_advect_markers(o.ptr, f); L.eo_refresh_marker_counts(o.ptr); L.eo_update_fluid_sources(o.ptr)
    L.eo_extrapolate(o.ptr, o.f32p(o.u), 1); L.eo_extrapolate(o.ptr, o.f32p(o.v), 2)
    L.eo_zero_bounds(o.ptr, o.f32p(o.u), 1); L.eo_zero_bounds(o.ptr, o.f32p(o.v), 2)
    L.eo_advect_u(o.ptr, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.utmp))
    L.eo_advect_v(o.ptr, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.vtmp))
    L.eo_apply_body_forces(o.ptr, o.f32p(o.vtmp), f)
    L.eo_zero_bounds(o.ptr, o.f32p(o.utmp), 1); L.eo_zero_bounds(o.ptr, o.f32p(o.vtmp), 2)
    L.eo_build_system(o.ptr, f, o.f32p(o.utmp), o.f32p(o.vtmp))
    return o, dt


class System:
    def __init__(self, o):
        self.o = o
        N = self.N = o.X
        fl = self.fl = o.count > 0
        self.n = int(fl.sum())
        idx = self.idx = -np.ones((N, N), dtype=np.int64)
        idx[fl] = np.arange(self.n)
        ys, xs = np.nonzero(fl)
        self.ys, self.xs = ys, xs
        rows, cols, vals = [np.arange(self.n)], [np.arange(self.n)], [o.a_diag[fl].astype(np.float64)]
        for dy, dx in ((0, 1), (1, 0), (0, -1), (-1, 0)):
            nb = idx[ys + dy, xs + dx]
            ok = nb >= 0
            rows.append(np.arange(self.n)[ok]); cols.append(nb[ok]); vals.append(-np.ones(int(ok.sum())))
        self.A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n, self.n))
        self.b = o.b[fl].copy()
        self.full_r = np.zeros((N, N))
        self.full_z = np.zeros((N, N))

    def tile(self, r):
        """M_tile^-1 r through the oracle (tile_records = 16, no coarse part)"""
        o = self.o
        cm = o.c.coarse_m
        o.c.coarse_m = 0
        self.full_r[self.fl] = r
        o.lib.eo_apply_preconditioner(o.ptr, o.f64p(self.full_r), o.f64p(self.full_z))
        o.c.coarse_m = cm
        return self.full_z[self.fl].copy()


def p_const(S, g):
    nx = (S.N + g - 1) // g
    c = (S.ys // g) * nx + S.xs // g
    return sp.csr_matrix((np.ones(S.n), (np.arange(S.n), c)), shape=(S.n, nx * nx)), nx


VERTEX = bool(int(os.environ.get("VERTEX", "0")))      # nodes AT cell centres (hats vanish at the neighbouring nodes: exact nine-point Galerkin stencils)
VOFF = int(os.environ.get("VOFF", "0"))                  # vertex form: level l + 1's node J sits on level l's node 2 J + VOFF


def p_bilinear(S, g):
    """nodes at the aggregates' centres; constant extrapolation beyond the outermost nodes"""
    nx = (S.N + g - 1) // g

    def w(c):
        xi = (c - g / 2) / g if VERTEX else (c + 0.5 - g / 2) / g
        j0 = np.floor(xi).astype(np.int64)
        f = xi - j0
        j1 = j0 + 1
        lo = j0 < 0
        j0 = np.where(lo, 0, j0); f = np.where(lo, 0.0, f)
        hi = j1 > nx - 1
        f = np.where(hi, 0.0, f); j1 = np.where(hi, nx - 1, j1)
        return j0, j1, f
    jx0, jx1, fx = w(S.xs)
    jy0, jy1, fy = w(S.ys)
    rows = np.tile(np.arange(S.n), 4)
    cols = np.concatenate([jy0 * nx + jx0, jy0 * nx + jx1, jy1 * nx + jx0, jy1 * nx + jx1])
    vals = np.concatenate([(1 - fy) * (1 - fx), (1 - fy) * fx, fy * (1 - fx), fy * fx])
    P = sp.csr_matrix((vals, (rows, cols)), shape=(S.n, nx * nx))
    P.sum_duplicates()
    return P, nx


def exact_solver(Ac):
    d = Ac.diagonal()
    act = d > 0
    ia = np.nonzero(act)[0]
    lu = spla.splu(sp.csc_matrix(Ac[ia][:, ia]))

    def solve(rc):
        x = np.zeros_like(rc)
        x[ia] = lu.solve(rc[ia])
        return x
    return solve


def agg2(nx):
    """2 x 2 piecewise-constant aggregation of an nx x nx node grid"""
    cx = (nx + 1) // 2
    I, J = np.divmod(np.arange(nx * nx), nx)
    return sp.csr_matrix((np.ones(nx * nx), (np.arange(nx * nx), (I // 2) * cx + J // 2)), shape=(nx * nx, cx * cx)), cx


def bil2(nx):
    """bilinear interpolation between node grids, coarse nodes at the centres of 2 x 2 fine nodes"""
    cx = (nx + 1) // 2

    def w(c):
        xi = (c - VOFF) / 2 if VERTEX else (c + 0.5 - 1.0) / 2
        j0 = np.floor(xi).astype(np.int64)
        f = xi - j0
        j1 = j0 + 1
        lo = j0 < 0
        j0 = np.where(lo, 0, j0); f = np.where(lo, 0.0, f)
        hi = j1 > cx - 1
        f = np.where(hi, 0.0, f); j1 = np.where(hi, cx - 1, j1)
        return j0, j1, f
    I, J = np.divmod(np.arange(nx * nx), nx)
    jx0, jx1, fx = w(J)
    jy0, jy1, fy = w(I)
    rows = np.tile(np.arange(nx * nx), 4)
    cols = np.concatenate([jy0 * cx + jx0, jy0 * cx + jx1, jy1 * cx + jx0, jy1 * cx + jx1])
    vals = np.concatenate([(1 - fy) * (1 - fx), (1 - fy) * fx, fy * (1 - fx), fy * fx])
    P = sp.csr_matrix((vals, (rows, cols)), shape=(nx * nx, cx * cx))
    P.sum_duplicates()
    return P, cx


def vcycle_solver(A0, nx0, top=256, omega=1.0, kappa=1.7, inter="agg", nsmooth=1, smoother="jacobi", theta=1.6):
    """the product's cycle: Jacobi from zero, restricted residual, recursion, scaled correction, Jacobi; dense top level"""
    levels = []
    A, nx = A0, nx0
    while nx * nx > top:
        P, cx = (agg2 if inter == "agg" else bil2)(nx)
        levels.append((A.tocsr(), P))
        A = (P.T @ A @ P).tocsr()
        if os.environ.get("SHOWNNZ"):
            print("   level nx=%d: max nnz per row %d" % (cx, int(np.diff(A.indptr).max())))
        nx = cx
    topsolve = exact_solver(A)

    def smooth(A, dinv, x, rhs, first):
        for k in range(nsmooth):
            if first and k == 0:
                x = omega * dinv * rhs
            else:
                x = x + omega * dinv * (rhs - A @ x)
        return x

    def cyc(l, rhs):
        if l == len(levels):
            return topsolve(rhs)
        A, P = levels[l]
        d = A.diagonal()
        dinv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0.0)
        if smoother == "l1":
            l1 = np.asarray(abs(A).sum(axis=1)).ravel()
            dinv = np.where(d > 0, 1.0 / np.where(l1 > 0, l1, 1), 0.0)
        if smoother == "gersh":      # per-node damping: omega_i = min(omega, theta / (1 + s_i)), s_i = sum |off-diagonals| / diagonal: the Gershgorin bound of D~^-1 A stays below theta < 2
            l1 = np.asarray(abs(A).sum(axis=1)).ravel()
            sden = np.where(d > 0, l1 / np.where(d > 0, d, 1), 1.0)      # 1 + s_i
            dinv = dinv * np.minimum(omega, theta / sden) / omega
        x = smooth(A, dinv, None, rhs, True)
        xc = cyc(l + 1, P.T @ (rhs - A @ x))
        x = x + kappa * (P @ xc)
        x = x * (d > 0)
        for k in range(nsmooth):
            x = x + omega * dinv * (rhs - A @ x)
        return x
    return lambda rc: cyc(0, rc), len(levels)


def pcg(S, coarse, tol=1e-6, maxit=3000, mult=False):
    """the reference's loop (main.c:742-766) with z = M_tile^-1 r + P C P^T r; mult: symmetric multiplicative (tile, coarse, tile)"""
    A, b = S.A, S.b

    def prec(r):
        if coarse is None:
            return S.tile(r)
        P, solve = coarse
        if not mult:
            return S.tile(r) + P @ solve(P.T @ r)
        z = S.tile(r)
        z = z + P @ solve(P.T @ (r - A @ z))
        return z + S.tile(r - A @ z)
    p = np.zeros_like(b)
    r = b.copy()
    z = prec(r)
    s = z.copy()
    sigma = z @ r
    hist = []
    for it in range(maxit):
        q = A @ s
        alpha = sigma / (q @ s)
        p += alpha * s
        r -= alpha * q
        res = np.abs(r).max()
        hist.append(res)
        if res <= tol:
            return it + 1, hist, p
        z = prec(r)
        sn = z @ r
        s = z + (sn / sigma) * s
        sigma = sn
    return maxit, hist, p


def main():
    N = int(sys.argv[1])
    wl = sys.argv[2] if len(sys.argv) > 2 else "half_tank"
    pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    want = sys.argv[4:]
    t0 = time.perf_counter()
    o, dt = state_before_projection(N, wl, pre)
    S = System(o)
    print("N=%d %s preroll=%d dt=%g fluid=%d max|b|=%g (%.1f s)" % (N, wl, pre, dt, S.n, np.abs(S.b).max(), time.perf_counter() - t0), flush=True)
    variants = {}

    def add(name, fn):
        if not want or any(name.startswith(w) for w in want):
            variants[name] = fn

    def v_const(g, ksc, exact=True, **kw):
        def make():
            P, nx = p_const(S, g)
            A0 = (P.T @ S.A @ P).tocsr()
            if exact:
                sol = exact_solver(A0)
                return (P, lambda rc: ksc * sol(rc)), {}
            sol, nl = vcycle_solver(A0, nx, **kw)
            return (P, sol), {}
        return make

    def v_bil(g, ksc, exact=True, **kw):
        def make():
            P, nx = p_bilinear(S, g)
            A0 = (P.T @ S.A @ P).tocsr()
            if exact:
                sol = exact_solver(A0)
                return (P, lambda rc: ksc * sol(rc)), {}
            sol, nl = vcycle_solver(A0, nx, **kw)
            return (P, sol), {}
        return make
    add("tile_only", lambda: (None, {}))
    add("const16_vcycle_product", v_const(16, 1.0, exact=False, omega=1.0, kappa=1.7))
    add("const16_exact_k1.0", v_const(16, 1.0))
    add("const16_exact_k1.7", v_const(16, 1.7))
    add("const8_exact_k1.0", v_const(8, 1.0))
    add("const8_exact_k1.5", v_const(8, 1.5))
    add("bil16_exact_k1.0", v_bil(16, 1.0))
    add("bil16_exact_k1.3", v_bil(16, 1.3))
    add("bil8_exact_k1.0", v_bil(8, 1.0))
    add("bil16_vcycle_agg_k1.7", v_bil(16, 1.0, exact=False, omega=0.8, kappa=1.7, inter="agg"))
    add("bil16_vcycle_bil_k1.0", v_bil(16, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil"))
    add("bil16_vcycle_bil_k1.0_s2", v_bil(16, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil", nsmooth=2))
    for om in (0.7, 0.8, 0.9, 1.0, 1.1):
        for ka in (1.0, 1.1, 1.2):
            add("scan_om%.1f_ka%.1f" % (om, ka), v_bil(16, 1.0, exact=False, omega=om, kappa=ka, inter="bil"))
    add("bil8_vcycle_bil", v_bil(8, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil"))
    add("bil4_vcycle_bil", v_bil(4, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil"))
    add("bil8_vcycle_gersh16", v_bil(8, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil", smoother="gersh", theta=1.6))
    add("bil8_vcycle_gersh18", v_bil(8, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil", smoother="gersh", theta=1.8))
    add("bil8_vcycle_l1", v_bil(8, 1.0, exact=False, omega=1.0, kappa=1.0, inter="bil", smoother="l1"))
    add("bil8_vcycle_bil_s2", v_bil(8, 1.0, exact=False, omega=0.8, kappa=1.0, inter="bil", nsmooth=2))
    add("mult_const16_exact_k1.0", v_const(16, 1.0))
    add("mult_bil16_exact_k1.0", v_bil(16, 1.0))
    for name, make in variants.items():
        t0 = time.perf_counter()
        coarse, _ = make()
        it, hist, p = pcg(S, coarse, mult=name.startswith("mult_"))
        print("%-34s %5d iterations  (res after 100: %.3g)  %.1f s" % (name, it, hist[min(99, len(hist) - 1)], time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
