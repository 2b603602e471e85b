"""NumPy prototype of the divide-and-conquer tridiagonal eigensolver that csrc/kernels_stedc.hip implements
(Cuppen's method with Gu/Eisenstat's vectors; LAPACK dlaed0-4 is the published algorithm it follows, with the secular
roots found by bisection on the bit pattern of the shift instead of dlaed4's rational interpolation).  Same steps, same
formulas, same order as the kernels: used to validate the numerics on degenerate / graded spectra before the port."""
import numpy as np, sys, struct

EPS = np.finfo(float).eps

def bits(x): return struct.unpack("<q", struct.pack("<d", x))[0]
def frombits(b): return struct.unpack("<d", struct.pack("<q", b))[0]

def secular_root(j, d, w2, rho):
    """root j of 1 + rho sum w2_i / (d_i - lam): returns (K, mu) with lam = d[K] + mu"""
    k = len(d)
    def g(K, mu):
        delta = (d - d[K]) - mu
        return 1.0 + rho * np.sum(w2 / delta)
    if j < k - 1:
        gap = d[j + 1] - d[j]
        half = 0.5 * gap
        # f at the midpoint, from the side of d_j
        fm = g(j, half)
        if fm >= 0:
            K, lo, hi, sign = j, 0.0, half, 1.0       # mu in (0, half], g increasing, g(lo+) = -inf
        else:
            K, lo, hi, sign = j + 1, 0.0, half, -1.0  # mu = -t, t in (0, half]; g(-t) decreasing in t; g(-half) < 0
    else:
        K, lo, hi, sign = k - 1, 0.0, rho * np.sum(w2), 1.0
        if hi == 0.0: return K, 0.0
    # bisection on the bit pattern of t = |mu| in (lo, hi]
    lb, hb = bits(0.0), bits(hi)
    # invariant: sign=+1: g(lo) < 0 <= g(hi);  sign=-1 (mu=-t): g(-lo) > 0 >= g(-hi)   (g(-hi) < 0 at start)
    while hb - lb > 1:
        mb = (lb + hb) // 2
        t = frombits(mb)
        val = g(K, sign * t)
        if sign > 0:
            if val >= 0: hb = mb
            else: lb = mb
        else:
            if val <= 0: hb = mb
            else: lb = mb
    return K, sign * frombits(hb)

def merge(D1, QT1, D2, QT2, e):
    """nodes (D ascending, QT rows = eigenvectors) joined by the off-diagonal e; returns D, QT of the union"""
    n1, n2 = len(D1), len(D2); N = n1 + n2
    rho = abs(e); sgn = 1.0 if e >= 0 else -1.0
    z = np.concatenate([QT1[:, n1 - 1], sgn * QT2[:, 0]]) / np.sqrt(2.0)
    rho = 2.0 * rho
    D = np.concatenate([D1, D2])
    QTold = np.zeros((N, N)); QTold[:n1, :n1] = QT1; QTold[n1:, n1:] = QT2
    perm = np.argsort(D, kind="stable")
    tol = 8.0 * EPS * max(np.abs(D).max(), np.abs(z).max())
    Dn = D.copy(); zn = z.copy()
    nd = []; defl = []; rots = []
    if rho * np.abs(z).max() <= tol:
        defl = list(perm)
    else:
        pj = -1
        for idx in perm:
            if rho * abs(zn[idx]) <= tol:
                defl.append(idx); continue
            if pj < 0:
                pj = idx; continue
            s = zn[pj]; c = zn[idx]
            tau = np.hypot(c, s)
            t = Dn[idx] - Dn[pj]
            c /= tau; s = -s / tau
            if abs(t * c * s) <= tol:
                zn[idx] = tau; zn[pj] = 0.0
                rots.append((pj, idx, c, s))
                tt = Dn[pj] * c * c + Dn[idx] * s * s
                Dn[idx] = Dn[pj] * s * s + Dn[idx] * c * c
                Dn[pj] = tt
                defl.append(pj)
                pj = idx
            else:
                nd.append(pj); pj = idx
        nd.append(pj)
    k = len(nd)
    U = np.zeros((N, N))
    newD = np.zeros(N)
    dl = Dn[nd] if k else np.zeros(0)
    wz = zn[nd] if k else np.zeros(0)
    dv = np.array([Dn[p] for p in defl]) if defl else np.zeros(0)
    lam = np.zeros(k); Ks = np.zeros(k, dtype=int); mus = np.zeros(k)
    for j in range(k):
        Ks[j], mus[j] = secular_root(j, dl, wz * wz, rho)
        lam[j] = dl[Ks[j]] + mus[j]
    # ranks in the merged ascending order
    rank_nd = np.array([j + np.sum(dv <= lam[j]) for j in range(k)], dtype=int)
    order_d = np.argsort(dv, kind="stable"); pos_d = np.empty(len(dv), dtype=int); pos_d[order_d] = np.arange(len(dv))
    rank_df = np.array([pos_d[p] + np.sum(lam < dv[p]) for p in range(len(dv))], dtype=int)
    if k:
        delta = (dl[:, None] - dl[Ks][None, :]) - mus[None, :]     # delta[i, j] = d_i - lam_j
        zhat = np.zeros(k)
        for i in range(k):
            prod = delta[i, i]
            for j in range(k):
                if j != i: prod *= delta[i, j] / (dl[i] - dl[j])
            zhat[i] = np.copysign(np.sqrt(abs(prod)), wz[i])
        for j in range(k):
            u = zhat / delta[:, j]
            u /= np.linalg.norm(u)
            U[nd, rank_nd[j]] = u
            newD[rank_nd[j]] = lam[j]
    for p in range(len(dv)):
        U[defl[p], rank_df[p]] = 1.0
        newD[rank_df[p]] = dv[p]
    for (a, b, c, s) in reversed(rots):
        ra = U[a].copy(); rb = U[b].copy()
        U[a] = c * ra - s * rb
        U[b] = s * ra + c * rb
    QTnew = U.T @ QTold
    return newD, QTnew

def stedc(d, e, leaf=32):
    n = len(d)
    d = np.array(d, dtype=float); e = np.array(e, dtype=float)
    scale = max(np.abs(d).max(), np.abs(e).max() if n > 1 else 0.0)
    if scale == 0: return np.zeros(n), np.eye(n)
    d /= scale; e /= scale
    bounds = list(range(0, n, leaf)) + [n]
    for b in bounds[1:-1]:
        r = abs(e[b - 1]); d[b - 1] -= r; d[b] -= r
    nodes = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        T = np.diag(d[a:b]) + np.diag(e[a:b - 1], 1) + np.diag(e[a:b - 1], -1)
        w, V = np.linalg.eigh(T)
        nodes.append((a, b, w, V.T.copy()))
    while len(nodes) > 1:
        nxt = []
        for i in range(0, len(nodes) - 1, 2):
            a, m, D1, Q1 = nodes[i]; _, b, D2, Q2 = nodes[i + 1]
            D, QT = merge(D1, Q1, D2, Q2, e[m - 1])
            nxt.append((a, b, D, QT))
        if len(nodes) % 2: nxt.append(nodes[-1])
        nodes = nxt
    return nodes[0][2] * scale, nodes[0][3].T

def check(name, d, e):
    n = len(d)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w, Z = stedc(d, e)
    wl = np.linalg.eigvalsh(T)
    sc = max(1e-300, np.abs(wl).max())
    print("%-28s n=%4d  |w-wl|/|w| %.2e  resid %.2e  orth %.2e  sorted %s" % (name, n, np.abs(np.sort(w) - wl).max() / sc,
          np.abs(T @ Z - Z * w).max() / sc, np.abs(Z.T @ Z - np.eye(n)).max(), bool(np.all(np.diff(w) >= 0))))

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    check("random", rng.standard_normal(n), rng.standard_normal(n - 1))
    check("1-2-1", 2 * np.ones(n), -np.ones(n - 1))
    check("wilkinson", np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1))
    check("graded", 10.0 ** (-np.arange(n) * 12.0 / n), 10.0 ** (-np.arange(n - 1) * 12.0 / n))
    check("zero offdiag blocks", rng.standard_normal(n), rng.standard_normal(n - 1) * (rng.random(n - 1) < 0.5))
    # tridiagonal of a matrix with a few eigenvalues of huge multiplicity (the path's generic elements)
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Dg = np.repeat(rng.standard_normal(6) * 3, n // 6 + 1)[:n]
    A = (Q0 * Dg) @ Q0.T; A = (A + A.T) / 2
    import scipy.linalg as sl
    H = sl.hessenberg(A)
    check("clustered (6 eigenvalues)", np.diag(H).copy(), np.diag(H, -1).copy())
    check("glued wilkinson", np.tile(np.abs(np.arange(21) - 10.0), n // 21 + 1)[:n], np.where((np.arange(n - 1) + 1) % 21 == 0, 1e-8, 1.0))
