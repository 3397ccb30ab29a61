"""Synthetic inputs of the BASELINE configs, for bench.py and tools/: plain numpy, no oracle.

The C2 "mapped-Cartesian" metric is the separable stretch s_a(x) = 1 + 0.3 sin(2 pi x / L_a + a):
Jg^{aa} = s_b s_c / s_a at a-face centres, Jinv = 1 / (s_0 s_1 s_2) at cell centres -- a diagonal, non-uniform
metric with the structure of the reference's CylindricalMap, handed to the solver as full coefficient arrays
exactly as LevelGeometry::getFCJgupPtr / getCCJinvPtr would (projection/AMRPressureSolver.cpp:163-463).
tests/test_synthetic.py checks these arrays against the oracle's generator bit for bit, so the timed problem and
the parity-tested problem are the same one.
"""
import numpy as np


def stretch_factor(a, x, L):
    return 1.0 + 0.3 * np.sin(2.0 * np.pi * x / L + a)


def stretched_diagonal_metric(lo, hi, dx, L, ndim=3):
    """Metric of one box [lo, hi] (cell indices, inclusive).
    -> ([Jg^{00} on x-faces, Jg^{11} on y-faces, Jg^{22} on z-faces] (None beyond ndim), Jinv), Fortran order;
    face array d has one more entry in direction d, face index i = low face of cell i."""
    def coords(face_dir):
        xs = []
        for d in range(3):
            n = hi[d] - lo[d] + 1 + (1 if d == face_dir else 0)
            idx = np.arange(lo[d], lo[d] + n, dtype=np.float64)
            xs.append((idx if d == face_dir else idx + 0.5) * dx[d])
        return xs

    def s(a, x):
        return stretch_factor(a, x, L[a]) if a < ndim else np.ones_like(x)

    def planes(x):
        return [s(0, x[0])[:, None, None], s(1, x[1])[None, :, None], s(2, x[2])[None, None, :]]

    jg = [None, None, None]
    for d in range(ndim):
        sv = planes(coords(d))
        num = np.ones(tuple(hi[q] - lo[q] + 1 + (1 if q == d else 0) for q in range(3)))
        for e in range(3):
            if e != d:
                num = num * sv[e]
        jg[d] = np.asfortranarray(num / sv[d])
    sv = planes(coords(-1))
    jinv = np.asfortranarray(1.0 / (sv[0] * sv[1] * sv[2]))
    return jg, jinv


def slab_partition(n, nparts, mode=None):
    """One box per GPU for the 512^3 problem.  mode "yz" (default; SOMAR_BENCH_PARTITION overrides): split z, then y; x is
    never split (x rows stay long: coalescing, and the marching kernels' tile columns divide a 512-wide box evenly).
    mode "xy": the horizontal-only 2-D grid of SURVEY.md 8(e) -- 1x2, 2x2, 2x4 boxes in (x, y), vertical columns whole (what a
    layout that also runs vertical line relaxation or the leptic solver needs).  Same halo surface per rank at N = 8
    (393 K cells) either way.  -> [(lo, hi)] in rank order."""
    import os
    mode = mode or os.environ.get("SOMAR_BENCH_PARTITION", "yz")
    if mode == "xy":
        split = {1: (1, 1, 1), 2: (1, 2, 1), 4: (2, 2, 1), 8: (2, 4, 1)}.get(nparts)
    elif mode == "yz":
        split = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (1, 2, 4)}.get(nparts)
    else:
        raise ValueError("partition mode must be 'yz' or 'xy'")
    if split is None:
        raise ValueError("number of parts must be 1, 2, 4 or 8")
    sz = [n // s for s in split]
    boxes = []
    for k in range(split[2]):
        for j in range(split[1]):
            for i in range(split[0]):
                lo = (i * sz[0], j * sz[1], k * sz[2])
                boxes.append((lo, tuple(a + b - 1 for a, b in zip(lo, sz))))
    return boxes


def boxes_of(lo, hi, bs):
    """[lo, hi] cut into boxes of bs cells, k slowest / i fastest (the order Chombo's domainSplit produces)"""
    out = []
    for k in range(lo[2], hi[2] + 1, bs[2]):
        for j in range(lo[1], hi[1] + 1, bs[1]):
            for i in range(lo[0], hi[0] + 1, bs[0]):
                out.append(((i, j, k), (min(i + bs[0], hi[0] + 1) - 1, min(j + bs[1], hi[1] + 1) - 1,
                                        min(k + bs[2], hi[2] + 1) - 1)))
    return out


def y_slab_owners(boxes, nranks):
    """Horizontal-only sharding of one level: boxes sorted by their y start are dealt to the ranks
    in contiguous, equally sized groups, so every rank owns one y-slab of EVERY level -- vertical columns stay
    whole, x rows stay long, and a fine box mostly sits over coarse cells of the same rank."""
    if nranks == 1:
        return [0] * len(boxes)
    ys = sorted({b[0][1] for b in boxes})
    rank_of_y = {y: min(q * nranks // len(ys), nranks - 1) for q, y in enumerate(ys)}
    return [rank_of_y[b[0][1]] for b in boxes]


def xy_block_owners(boxes, nranks):
    """Horizontal 2-D sharding of one level: the level's bounding rectangle in (x, y) is cut into a px x py grid of blocks
    (1x2, 2x2, 2x4 for 2, 4, 8 ranks; y gets the larger factor, x rows stay as long as possible) and a box belongs to the block
    its low corner falls in.  Against y-slabs at N = 8 a rank's halo shrinks (C4's finest level: 262 K -> 196 K cells per
    exchange) and it gains x-neighbours.  Vertical columns stay whole."""
    if nranks == 1:
        return [0] * len(boxes)
    px, py = {2: (1, 2), 4: (2, 2), 8: (2, 4)}.get(nranks, (1, nranks))
    x0, x1 = min(b[0][0] for b in boxes), max(b[1][0] for b in boxes) + 1
    y0, y1 = min(b[0][1] for b in boxes), max(b[1][1] for b in boxes) + 1
    out = []
    for lo, hi in boxes:
        bx = min((lo[0] - x0) * px // (x1 - x0), px - 1)
        by = min((lo[1] - y0) * py // (y1 - y0), py - 1)
        out.append(by * px + bx)
    return out


def level_owners(boxes, nranks, mode=None):
    """owners of one level's boxes: "slab" (y-slabs, default; SOMAR_BENCH_OWNERS overrides) or "block" (x-y blocks)"""
    import os
    mode = mode or os.environ.get("SOMAR_BENCH_OWNERS", "slab")
    if mode == "block":
        return xy_block_owners(boxes, nranks)
    if mode != "slab":
        raise ValueError("owners mode must be 'slab' or 'block'")
    return y_slab_owners(boxes, nranks)


def lockexchange_hierarchy(config="c3", scale=1, box=128, mult=4, nranks=1):
    """Box layouts of the LockExchange-shaped BASELINE configs (the refined regions are this
    repository's choice, recorded in BASELINE.md):
      c3   512x512x64 base, L = (15,3,2), y periodic, level 1 = (2,2,1) refinement of the central half in x
      c4   1024x1024x128 base + two (2,2,1) levels (central half, central quarter in x)
      le3d exec/inputs.LockExchange_Cartesian3D.machine: 64x96x64 times mult, one level refined by (4,1,1)
      le2d exec/inputs.LockExchange_Cartesian2D.machine: 2-D, 128x64 times mult, L = (15,2), one level (4,1)
    scale divides every extent.  -> dict(n0, L, periodic, ratios, levels, owners, flat)"""
    s = scale
    if config == "c3":
        n0, nlev = (512 // s, 512 // s, 64 // s), 2
    elif config == "le3d":
        n0, nlev = (64 * mult // s, 96 * mult // s, 64 * mult // s), 2
    elif config == "le2d":
        n0, nlev = (128 * mult // s, 64 * mult // s, 1), 2
    elif config == "c4":
        n0, nlev = (1024 // s, 1024 // s, 128 // s), 3
    else:
        raise ValueError(config)
    L = (15.0, 3.0, 2.0)
    le = config in ("le3d", "le2d")
    ratios = [(4, 1, 1)] * (nlev - 1) if le else [(2, 2, 1)] * (nlev - 1)
    flat = config == "le2d"
    if flat:
        L = (15.0, 2.0, 1.0)
    bs = (max(box // s, 8), max(box // s, 8), n0[2])
    levels = [boxes_of((0, 0, 0), tuple(a - 1 for a in n0), bs)]
    n = list(n0)
    frac = 2
    for l in range(1, nlev):
        r = ratios[l - 1]
        n = [n[0] * r[0], n[1] * r[1], n[2] * r[2]]
        w = n[0] // frac          # central half, then central quarter (of the refined index space)
        lo_x = (n[0] - w) // 2
        lo_x -= lo_x % ((1 if le else 2) * bs[0])
        levels.append(boxes_of((lo_x, 0, 0), (lo_x + w - 1, n[1] - 1, n[2] - 1), bs))
        frac *= 2
    return {"n0": n0, "L": L, "periodic": (False, False, False) if flat else (False, True, False), "ratios": ratios,
            "levels": levels, "owners": [level_owners(b, nranks) for b in levels], "flat": flat,
            "dx0": tuple(L[d] / n0[d] for d in range(3))}


def terrain_metric(lo, hi, dx, L):
    """BASELINE config C5's terrain-following map, the form of geometry/BathymetricBaseMapF.ChF:85-110:
    x = xi, y = eta, z = d(xi, eta) + (1 - d/H) zeta over the depth h = H - d = H s,  s = 0.5 + 0.3 exp(-r^2 / w^2),
    r^2 = (x - L0/2)^2 + (y - L1/2)^2, w = min(L0, L1) / 4, H = L2  =>  J = z_zeta = s, z_xi = s_x (zeta - H), z_eta = s_y (zeta - H),
        J g^{xi b}   = (s, 0, -z_xi)            on xi-faces
        J g^{eta b}  = (0, s, -z_eta)           on eta-faces
        J g^{zeta b} = (-z_xi, -z_eta, (1 + z_xi^2 + z_eta^2) / s)   on zeta-faces,      Jinv = 1 / s at cell centres.
    Box [lo, hi] (cell indices, inclusive).  -> ([jg0, jg1, jg2] each (faces shape) + (3,) Fortran-ordered = component
    slowest, the FluxBox layout of LevelGeometry::getFCJgupPtr; Jinv)"""
    H = L[2]
    w = min(L[0], L[1]) / 4.0

    def coords(face_dir):
        xs = []
        for d in range(3):
            n = hi[d] - lo[d] + 1 + (1 if d == face_dir else 0)
            idx = np.arange(lo[d], lo[d] + n, dtype=np.float64)
            xs.append((idx if d == face_dir else idx + 0.5) * dx[d])
        return np.meshgrid(*xs, indexing="ij")

    def fields(X):
        ex = np.exp(-((X[0] - 0.5 * L[0]) ** 2 + (X[1] - 0.5 * L[1]) ** 2) / (w * w))
        s = 0.5 + 0.3 * ex
        sx = -0.3 * ex * (2.0 * (X[0] - 0.5 * L[0]) / (w * w))
        sy = -0.3 * ex * (2.0 * (X[1] - 0.5 * L[1]) / (w * w))
        zx = sx * (X[2] - H)
        zy = sy * (X[2] - H)
        return s, zx, zy

    jg = []
    for d in range(3):
        s, zx, zy = fields(coords(d))
        a = np.zeros(s.shape + (3,), order="F")
        if d == 0:
            a[..., 0], a[..., 2] = s, -zx
        elif d == 1:
            a[..., 1], a[..., 2] = s, -zy
        else:
            a[..., 0], a[..., 1], a[..., 2] = -zx, -zy, (1.0 + zx * zx + zy * zy) / s
        jg.append(a)
    s, _, _ = fields(coords(-1))
    return jg, np.asfortranarray(1.0 / s)


def terrain_nodal_depth(nlo, nhi, dx, L):
    """The NODAL depth field d(x, y) = H (1 - s) of terrain_metric's map, on nodes [nlo, nhi] (inclusive) of a level with
    spacing dx: what a BathymetricBaseMap subclass's fill_bathymetry returns, the input of somar_solver_set_metric_map."""
    H = L[2]
    w = min(L[0], L[1]) / 4.0
    x = (np.arange(nlo[0], nhi[0] + 1, dtype=np.float64) * dx[0])[:, None]
    y = (np.arange(nlo[1], nhi[1] + 1, dtype=np.float64) * dx[1])[None, :]
    s = 0.5 + 0.3 * np.exp(-((x - 0.5 * L[0]) ** 2 + (y - 0.5 * L[1]) ** 2) / (w * w))
    return np.asfortranarray(H * (1.0 - s))


def c5_hierarchy(scale=1, box=64, nranks=1, nlev=4):
    """BASELINE config C5's shape: terrain-following NON-diagonal metric (terrain_metric), 4 levels each
    refined by (2,2,1), nested around the topographic bump at the domain centre: level l covers the central 2^-l of the
    horizontal extent in x and y.  Base 512x512x64 / scale over L = (8, 8, 1), no periodic direction, boxes of
    box x box x nz cells (columns stay whole), every level's boxes in y-slabs over the ranks.  Each level has
    (512/scale)^2 x 64/scale cells.  -> the dict lockexchange_hierarchy returns, plus "metric": "terrain"."""
    s = scale
    n0 = (512 // s, 512 // s, 64 // s)
    L = (8.0, 8.0, 1.0)
    ratios = [(2, 2, 1)] * (nlev - 1)
    bs = (max(box // s, 8), max(box // s, 8), n0[2])
    levels = [boxes_of((0, 0, 0), tuple(a - 1 for a in n0), bs)]
    n = list(n0)
    for l in range(1, nlev):
        n = [n[0] * 2, n[1] * 2, n[2]]
        w = [n[0] // (2 ** l), n[1] // (2 ** l)]          # the central 2^-l of the level-l index space
        lo = [(n[0] - w[0]) // 2, (n[1] - w[1]) // 2]
        lo = [a - a % (2 * bs[0]) for a in lo]
        levels.append(boxes_of((lo[0], lo[1], 0), (lo[0] + w[0] - 1, lo[1] + w[1] - 1, n[2] - 1), bs))
    return {"n0": n0, "L": L, "periodic": (False, False, False), "ratios": ratios, "levels": levels,
            "owners": [level_owners(b, nranks) for b in levels], "flat": False, "metric": "terrain",
            "dx0": tuple(L[d] / n0[d] for d in range(3))}
