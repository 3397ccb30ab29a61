"""The synthetic inputs bench.py and tools/ feed the HIP library (somar_amd/synthetic.py, no oracle import) are the
same arrays the parity tests feed both sides (the oracle's generator): bit for bit.  And the measured path of
bench.py / tools/ stays clear of oracle/: only bench.py's cpu_baseline leg may touch it."""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stretched_metric_equals_the_oracles_generator(oracle):
    so = oracle
    from somar_amd import synthetic
    n = (16, 12, 8)
    L = (1.0, 2.0, 0.5)
    dx = tuple(L[d] / n[d] for d in range(3))
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (8, 4, 8))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    for gi, g in enumerate(grids):
        jg, jinv = synthetic.stretched_diagonal_metric(g.lo, g.hi, dx, L)
        for d in range(3):
            np.testing.assert_array_equal(jg[d], Jgup[gi][d].a[..., d])
        np.testing.assert_array_equal(jinv, Jinv[gi].a[..., 0])


def test_terrain_metric_equals_the_oracles_generator(oracle):
    so = oracle
    from somar_amd import synthetic
    n = (16, 12, 8)
    L = (4.0, 3.0, 0.5)
    dx = tuple(L[d] / n[d] for d in range(3))
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (8, 4, 8))
    Jgup, Jinv = so.make_terrain_metric(grids, dx, L, dom)
    for gi, g in enumerate(grids):
        jg, jinv = synthetic.terrain_metric(g.lo, g.hi, dx, L)
        for d in range(3):
            np.testing.assert_array_equal(jg[d], Jgup[gi][d].a)
        np.testing.assert_array_equal(jinv, Jinv[gi].a[..., 0])
        assert np.all(jinv > 0) and float(np.abs(jg[2][..., 0]).max()) > 0.0   # positive Jacobian, genuinely non-diagonal


def test_slab_partition_tiles_the_domain():
    from somar_amd import synthetic
    for parts in (1, 2, 4, 8):
        boxes = synthetic.slab_partition(512, parts)
        assert len(boxes) == parts
        cells = 0
        for lo, hi in boxes:
            assert lo[0] == 0 and hi[0] == 511          # x rows are never split
            cells += int(np.prod([h - a + 1 for a, h in zip(lo, hi)]))
        assert cells == 512 ** 3
        assert len({b for b in boxes}) == parts


def test_xy_partition_and_block_owners():
    """the horizontal-only 2-D alternatives: whole vertical columns, every rank the same number of cells"""
    from somar_amd import synthetic
    for parts in (1, 2, 4, 8):
        boxes = synthetic.slab_partition(512, parts, mode="xy")
        assert len(boxes) == parts and len(set(boxes)) == parts
        assert all(lo[2] == 0 and hi[2] == 511 for lo, hi in boxes)
        assert sum(int(np.prod([h - a + 1 for a, h in zip(lo, hi)])) for lo, hi in boxes) == 512 ** 3
    H = synthetic.lockexchange_hierarchy("c4", 1, 128, 4, 8)
    for boxes in H["levels"]:
        own = synthetic.xy_block_owners(boxes, 8)
        per = np.bincount(own, minlength=8)
        assert per.min() == per.max() and per.sum() == len(boxes)      # balanced: the same number of 128^3 boxes everywhere
        # a rank's boxes form one rectangle in (x, y)
        for r in range(8):
            mine = [b for b, o in zip(boxes, own) if o == r]
            xs, ys = sorted({b[0][0] for b in mine}), sorted({b[0][1] for b in mine})
            assert len(mine) == len(xs) * len(ys)
    assert synthetic.level_owners(H["levels"][0], 8, "slab") == synthetic.y_slab_owners(H["levels"][0], 8)


def test_mt19937_64_field_is_the_standard_generator():
    """somar_host_fill_mt19937_64 = std::mt19937_64 + std::uniform_real_distribution (host code, no GPU): the C++ standard pins the
    10000th output of a default-seeded (5489) mt19937_64 to 9981545732273789042; libstdc++ maps one 64-bit draw to [0, 1) by
    dividing by 2^64"""
    from somar_amd import api
    a = api.host_random_field((100, 100, 1), 5489, 0.0, 1.0)
    assert abs(a.ravel(order="F")[9999] * 2.0 ** 64 - 9981545732273789042) <= 4096      # one ulp of 2^63 is 2048
    b = api.host_random_field((4, 3, 2), 12345)
    assert b.min() >= -1.0 and b.max() < 1.0 and b.flags.f_contiguous
    np.testing.assert_array_equal(b, api.host_random_field((4, 3, 2), 12345))


def _oracle_imports_outside(path, allowed_functions):
    """names of the scopes in which `path` imports anything from oracle/, minus the allowed ones"""
    tree = ast.parse(open(path).read())
    bad = []

    def visit(node, scope):
        for child in ast.iter_child_nodes(node):
            s = child.name if isinstance(child, (ast.FunctionDef, ast.ClassDef)) else scope
            if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                bad.append(scope)
            if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                bad.append(scope)
            visit(child, s)
    visit(tree, "<module>")
    return [b for b in bad if b not in allowed_functions]


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    assert _oracle_imports_outside(os.path.join(ROOT, "bench.py"), {"cpu_baseline"}) == []
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            assert _oracle_imports_outside(os.path.join(ROOT, "tools", f), set()) == [], f
