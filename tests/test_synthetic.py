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
