"""Seeded synthetic inputs of the shapes the benchmarks name (NumPy, host side).

SNAP .mtx files are not available offline, so tests and fixtures use graphs
generated here.  Every generator returns zero-based (row, col) int32 arrays
sorted by (col, row) with duplicates removed, which is the order SuiteSparse
.mtx files list their entries in; the reference loader
(serial_newblock_clock.cpp:108-112) keeps that order inside each CSR row, so
rows come out with ascending column ids.
"""
import numpy as np


def _finish(row, col, n):
    key = np.unique(col.astype(np.int64) * n + row.astype(np.int64))
    return (key % n).astype(np.int32), (key // n).astype(np.int32)


def uniform(n, draws, seed):
    """`draws` uniform (row, col) pairs on an n x n grid."""
    rng = np.random.default_rng(seed)
    row = rng.integers(0, n, draws)
    col = rng.integers(0, n, draws)
    return _finish(row, col, n)


def rmat(scale, edge_factor, seed, diagonal=False, a=0.57, b=0.19, c=0.19):
    """Graph500 R-MAT: edge_factor * 2^scale draws, one uniform number per level
    choosing the quadrant: level l decides bit l of both ids (level 0 = least
    significant bit); row bit set iff u >= a+b, column bit set iff a <= u < a+b or
    u >= a+b+c.  No vertex scrambling.
    `diagonal=True` adds (i, i) for every i (no empty rows)."""
    n = 1 << scale
    m = edge_factor << scale
    rng = np.random.default_rng(seed)
    row = np.zeros(m, np.int64)
    col = np.zeros(m, np.int64)
    for level in range(scale):
        u = rng.random(m)
        rbit = u >= a + b
        cbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
        row |= rbit.astype(np.int64) << level
        col |= cbit.astype(np.int64) << level
    if diagonal:
        d = np.arange(n, dtype=np.int64)
        row = np.concatenate([row, d])
        col = np.concatenate([col, d])
    return _finish(row, col, n)


def web_like(n=916428, m=5105039, seed=3):
    """A graph of the size and rough shape of SNAP web-Google (916 428 pages, 5.1 M links; the file itself is not
    available offline): geometric out-degrees scaled to m links, three targets in ten drawn from a Pareto-ranked
    popular head, the others uniform."""
    rng = np.random.default_rng(seed)
    deg = rng.geometric(0.18, n)
    deg = (deg * (m / deg.sum())).astype(np.int64) + 1
    row = np.repeat(np.arange(n), deg)[:m]
    pop = (rng.pareto(1.2, row.size) * 50).astype(np.int64) % n
    col = np.where(rng.random(row.size) < 0.3, pop, rng.integers(0, n, row.size))
    return _finish(row, col, n)


def stanford_like(n=281903, m=2312497, seed=8):
    """A graph of the size and rough shape of SNAP web-Stanford (281 903 pages, 2.3 M links, out-degrees up to 255,
    in-degrees up to tens of thousands): ordinary pages with geometric out-degrees whose targets are a Pareto-ranked
    popular head (four in ten) or uniform, plus 3 000 navigation pages in 60 sites of 50 that link to every other
    navigation page of their site and to 100-200 pages more, so that rows of A*A reach tens of thousands of products
    (the window bins), as the link-dense cores of real web graphs make them."""
    rng = np.random.default_rng(seed)
    nav = rng.choice(n, 3000, replace=False).reshape(60, 50)
    rows, cols = [], []
    for site in nav:
        rows.append(np.repeat(site, 50))
        cols.append(np.tile(site, 50))
        extra = rng.integers(100, 200, 50)
        rows.append(np.repeat(site, extra))
        cols.append(rng.integers(0, n, int(extra.sum())))
    used = sum(r.size for r in rows)
    deg = rng.geometric(0.14, n)
    deg = (deg * ((m - used) / deg.sum())).astype(np.int64) + 1
    row = np.repeat(np.arange(n), deg)[:m - used]
    pop = (rng.pareto(1.0, row.size) * 20).astype(np.int64) % n
    navt = nav.reshape(-1)[rng.integers(0, 3000, row.size)]
    u = rng.random(row.size)
    col = np.where(u < 0.35, pop, np.where(u < 0.45, navt, rng.integers(0, n, row.size)))
    rows.append(row)
    cols.append(col)
    return _finish(np.concatenate(rows), np.concatenate(cols), n)


def askubuntu_like(n=159316, m=964437, seed=5):
    """A graph of the size and rough shape of SNAP sx-askubuntu (159 316 users, 964 437 timed interactions, about
    0.6 M distinct pairs): both ends of an interaction drawn from heavy-tailed activity ranks (user id = n*u^3 for the
    one who writes, n*u^4 for the one written to: a few users answer or are answered thousands of times), repeated
    pairs collapse as in the static graph."""
    rng = np.random.default_rng(seed)
    src = (n * rng.random(m) ** 3).astype(np.int64)
    dst = (n * rng.random(m) ** 4).astype(np.int64)
    return _finish(src, dst, n)


def write_mtx(path, n, row, col, pattern=True):
    """MatrixMarket coordinate file, 1-based, in the given entry order."""
    with open(path, "w") as f:
        kind = "pattern" if pattern else "real"
        f.write(f"%%MatrixMarket matrix coordinate {kind} general\n")
        f.write(f"{n} {n} {len(row)}\n")
        if pattern:
            np.savetxt(f, np.stack([row + 1, col + 1], 1), fmt="%d %d")
        else:
            np.savetxt(f, np.stack([row + 1, col + 1, np.ones_like(row)], 1), fmt="%d %d %d")
