"""Fill of cluster-pair tiles at the headline box (CPU, numpy/scipy): how many lane-pairs a tile kernel would
evaluate for the P listed pairs, for several cluster sizes and clusterings.  Round-5 item 1(a) go / no-go input."""
import sys, numpy as np
sys.path.insert(0, ".")
from scipy.spatial import cKDTree
import importlib
W = importlib.import_module("torchpme_amd.workloads")

def box(name):
    # positions only (avoid the 15 s host list): replicate the generator's positions through its own function with cutoff tiny
    f = getattr(W, name)
    w = f(cutoff=0.5) if name != "dispersion_box" else f(cutoff=0.5)
    return w.positions, w.cell[0, 0]

def clusters_grid(pos, L, c):
    """cells of side ~ (c/rho)^(1/3), atoms sorted by cell (z fastest), chunked in groups of c inside the raster order"""
    N = len(pos)
    rho = N / L**3
    for scale in (1.0,):
        nc = max(1, int(round(L / (c / rho) ** (1 / 3) * scale)))
        ijk = np.floor((pos % L) / L * nc).astype(int) % nc
        key = (ijk[:, 0] * nc + ijk[:, 1]) * nc + ijk[:, 2]
        order = np.argsort(key, kind="stable")
    cl = np.empty(N, int); cl[order] = np.arange(N) // c
    return cl

def clusters_bisect(pos, L, c):
    """recursive median bisection along the longest extent until groups of c (compact, equal-sized)"""
    N = len(pos)
    cl = np.empty(N, int)
    nxt = [0]
    def rec(idx):
        if len(idx) <= c:
            cl[idx] = nxt[0]; nxt[0] += 1; return
        p = pos[idx]
        ext = p.max(0) - p.min(0)
        d = int(np.argmax(ext))
        o = np.argsort(p[:, d], kind="stable")
        half = (len(idx) // 2 + c - 1) // c * c if len(idx) > 2 * c else c
        half = min(half, len(idx) - 1)
        rec(idx[o[:half]]); rec(idx[o[half:]])
    rec(np.arange(N))
    return cl

def report(pos, L, rc, c, cl, pairs):
    i, j = pairs[:, 0], pairs[:, 1]
    ci, cj = cl[i], cl[j]
    a, b = np.minimum(ci, cj), np.maximum(ci, cj)
    ncl = cl.max() + 1
    key = a * ncl + b
    uk = np.unique(key)
    diag = (uk // ncl == uk % ncl).sum()
    ntile = len(uk)
    lane_pairs = ntile * c * c
    print(f"  c={c}: clusters {ncl}, tiles {ntile} (diag {diag}), lane-pairs {lane_pairs/1e6:.2f} M for P={len(pairs)/1e6:.2f} M "
          f"-> fill {len(pairs)/lane_pairs:.3f}; lane-pairs / 2P = {lane_pairs/(2*len(pairs)):.3f}")

for name, rc in (("water_box", 9.0), ("dispersion_box", 9.0)):
    if name == "dispersion_box" and len(sys.argv) < 2: continue
    pos, L = box(name)
    print(name, len(pos), L)
    t = cKDTree(pos % L, boxsize=L)
    pairs = t.query_pairs(rc, output_type="ndarray")
    print("  pairs", len(pairs))
    for c in (2, 3, 4, 8):
        report(pos, L, rc, c, clusters_grid(pos, L, c), pairs)
        report(pos, L, rc, c, clusters_bisect(pos % L, L, c), pairs)
