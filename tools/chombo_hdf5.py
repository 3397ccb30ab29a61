#!/usr/bin/env python3
"""Chombo-style HDF5 level dumps for the solver inspector, written through ctypes on the HDF5 C library
-- h5py is not in the image, libhdf5 is (/opt/conda/lib/libhdf5.so); where neither is found the callers fall back to .npz.

What the reference writes: OutputMappedAMRMultiGridInspector -> MappedAMRMultiGrid::outputAMR -> MappedAMRPoissonOp::outputAMR
(calculus/AMRElliptic/MappedAMRMultiGrid.H:305-362, MappedAMRPoissonOp.cpp:2157-2186) -> WriteAnisotropicAMRHierarchyHDF5
(utils/Printing.cpp:736-827): a header (filetype "VanillaAMRFileType", num_levels, num_components, max_level, time,
component_N = "comp_%03d"), then Chombo's writeLevel per level with the level's domain, dx (a RealVect), dt, time, the
refinement ratio to the next finer level (an IntVect) and the data's ghost vector.

writeLevel itself is Chombo 3.1 (EXTERNAL, not under /root/reference); this file restates the plot-file layout Chombo
publishes (and VisIt / ChomboVis read):
    /                        attributes: the header above
    /Chombo_global           attributes: SpaceDim (int), testReal (double)
    /level_N                 attributes: dx (double) and vec_dx (realvect: x, y, z), dt, time, ref_ratio (int) and
                             vec_ref_ratio (intvect: intvecti, intvectj, intvectk), prob_domain (box: lo_i .. hi_k)
    /level_N/boxes           one compound {lo_i, lo_j, lo_k, hi_i, hi_j, hi_k} per box
    /level_N/data:datatype=0 doubles: box after box, component slowest, Fortran order inside a box (grown by outputGhost)
    /level_N/data:offsets=0  int64, nboxes + 1
    /level_N/Processors      int per box
    /level_N/data_attributes attributes: comps, ghost (intvect), outputGhost (intvect), objectType "FArrayBox"
LAYOUT UNPINNED: the reference ships no HDF5 file to compare with; tests/test_chombo_hdf5.py checks the file against this
description by reading it back through the same library."""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
_LIB = None


def lib():
    """the HDF5 C library, or None"""
    global _LIB
    if _LIB is not None:
        return _LIB or None
    cands = [os.environ.get("SOMAR_HDF5_LIB"), "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5")]
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            L.H5open.restype = C.c_int
            if L.H5open() < 0:
                continue
        except OSError:
            continue
        for name, res, args in [
                ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
                ("H5Fclose", C.c_int, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
                ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Gclose", C.c_int, [hid_t]),
                ("H5Screate", hid_t, [C.c_int]), ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
                ("H5Sclose", C.c_int, [hid_t]), ("H5Sget_simple_extent_npoints", C.c_int64, [hid_t]),
                ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tcreate", hid_t, [C.c_int, C.c_size_t]),
                ("H5Tinsert", C.c_int, [hid_t, C.c_char_p, C.c_size_t, hid_t]), ("H5Tclose", C.c_int, [hid_t]),
                ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]),
                ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]),
                ("H5Aclose", C.c_int, [hid_t]),
                ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dget_space", hid_t, [hid_t]),
                ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dclose", C.c_int, [hid_t])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _LIB = L
        return L
    _LIB = False
    return None


def _t(name):
    return hid_t.in_dll(lib(), name).value


H5F_ACC_TRUNC, H5F_ACC_RDONLY, H5P_DEFAULT, H5S_SCALAR, H5T_COMPOUND, H5S_ALL = 2, 0, 0, 0, 6, 0


class _Writer:
    def __init__(self, path):
        self.L = lib()
        if self.L is None:
            raise RuntimeError("no HDF5 library (set SOMAR_HDF5_LIB)")
        self.f = self.L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        if self.f < 0:
            raise RuntimeError("H5Fcreate failed: " + path)
        self.INT, self.DBL, self.I64 = _t("H5T_NATIVE_INT_g"), _t("H5T_NATIVE_DOUBLE_g"), _t("H5T_NATIVE_LLONG_g")
        L = self.L
        self.intvect = L.H5Tcreate(H5T_COMPOUND, 12)
        for q, n in enumerate(("intvecti", "intvectj", "intvectk")):
            L.H5Tinsert(self.intvect, n.encode(), 4 * q, self.INT)
        self.realvect = L.H5Tcreate(H5T_COMPOUND, 24)
        for q, n in enumerate(("x", "y", "z")):
            L.H5Tinsert(self.realvect, n.encode(), 8 * q, self.DBL)
        self.box = L.H5Tcreate(H5T_COMPOUND, 24)
        for q, n in enumerate(("lo_i", "lo_j", "lo_k", "hi_i", "hi_j", "hi_k")):
            L.H5Tinsert(self.box, n.encode(), 4 * q, self.INT)

    def close(self):
        for t in (self.intvect, self.realvect, self.box):
            self.L.H5Tclose(t)
        self.L.H5Fclose(self.f)

    def group(self, parent, name):
        g = self.L.H5Gcreate2(parent, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert g >= 0
        return g

    def attr(self, obj, name, typ, buf):
        L = self.L
        s = L.H5Screate(H5S_SCALAR)
        a = L.H5Acreate2(obj, name.encode(), typ, s, H5P_DEFAULT, H5P_DEFAULT)
        assert a >= 0, name
        assert L.H5Awrite(a, typ, C.cast(buf, C.c_void_p)) >= 0
        L.H5Aclose(a)
        L.H5Sclose(s)

    def attr_int(self, obj, name, v):
        self.attr(obj, name, self.INT, C.byref(C.c_int(int(v))))

    def attr_real(self, obj, name, v):
        self.attr(obj, name, self.DBL, C.byref(C.c_double(float(v))))

    def attr_str(self, obj, name, v):
        L = self.L
        b = v.encode()
        t = L.H5Tcopy(_t("H5T_C_S1_g"))
        L.H5Tset_size(t, max(len(b), 1))
        self.attr(obj, name, t, C.create_string_buffer(b, max(len(b), 1)))
        L.H5Tclose(t)

    def attr_intvect(self, obj, name, v):
        self.attr(obj, name, self.intvect, (C.c_int * 3)(*[int(x) for x in v]))

    def attr_realvect(self, obj, name, v):
        self.attr(obj, name, self.realvect, (C.c_double * 3)(*[float(x) for x in v]))

    def attr_box(self, obj, name, lo, hi):
        self.attr(obj, name, self.box, (C.c_int * 6)(*[int(x) for x in tuple(lo) + tuple(hi)]))

    def dataset(self, parent, name, typ, arr, count=None):
        L = self.L
        n = hsize_t(len(arr) if count is None else count)
        s = L.H5Screate_simple(1, C.byref(n), None)
        d = L.H5Dcreate2(parent, name.encode(), typ, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert d >= 0, name
        if n.value:
            assert L.H5Dwrite(d, typ, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Dclose(d)
        L.H5Sclose(s)


def write_hierarchy(path, levels, domain, dx, ref_ratios, names=None, time=0.0, dt=0.0, ghost=(0, 0, 0)):
    """WriteAnisotropicAMRHierarchyHDF5 (utils/Printing.cpp:762-827).
    levels: per level a list of (lo, hi, array) -- array shaped (nx, ny, nz[, ncomp]) over the box grown by `ghost`;
    domain: (lo, hi) of level 0; dx: level-0 spacing (3 values); ref_ratios: per level -> next finer level, 3 ints each."""
    nlev = len(levels)
    first = np.asarray(levels[0][0][2])
    ncomp = first.shape[3] if first.ndim == 4 else 1
    names = names or ["comp_%03d" % c for c in range(ncomp)]     # MappedAMRPoissonOp::outputAMR's names
    W = _Writer(path)
    try:
        root = W.L.H5Gopen2(W.f, b"/", H5P_DEFAULT)
        W.attr_str(root, "filetype", "VanillaAMRFileType")
        W.attr_int(root, "num_levels", nlev)
        W.attr_int(root, "num_components", ncomp)
        W.attr_int(root, "max_level", nlev - 1)
        W.attr_real(root, "time", time)
        for c, nm in enumerate(names):
            W.attr_str(root, "component_%d" % c, nm)
        W.L.H5Gclose(root)
        g = W.group(W.f, "Chombo_global")
        W.attr_int(g, "SpaceDim", 3)
        W.attr_real(g, "testReal", 0.0)
        W.L.H5Gclose(g)
        dlo, dhi = [list(x) for x in domain]
        dxl, dtl = [float(x) for x in dx], float(dt)
        for l, boxes in enumerate(levels):
            ref = tuple(ref_ratios[l]) if l < nlev - 1 else (1, 1, 1)
            if l > 0:
                r = ref_ratios[l - 1]
                dlo = [a * b for a, b in zip(dlo, r)]
                dhi = [(a + 1) * b - 1 for a, b in zip(dhi, r)]
                dtl /= r[0]                                        # "HACK - just use 0 dir ref ratio" (Printing.cpp:808)
                dxl = [a / b for a, b in zip(dxl, r)]
            g = W.group(W.f, "level_%d" % l)
            W.attr_real(g, "dx", dxl[0])
            W.attr_realvect(g, "vec_dx", dxl)
            W.attr_real(g, "dt", dtl)
            W.attr_real(g, "time", time)
            W.attr_int(g, "ref_ratio", ref[0])
            W.attr_intvect(g, "vec_ref_ratio", ref)
            W.attr_box(g, "prob_domain", dlo, dhi)
            bx = np.zeros((len(boxes), 6), dtype=np.int32)
            offs = np.zeros(len(boxes) + 1, dtype=np.int64)
            chunks = []
            for q, (lo, hi, arr) in enumerate(boxes):
                bx[q] = list(lo) + list(hi)
                a = np.asarray(arr, dtype=np.float64)
                if a.ndim == 3:
                    a = a[..., None]
                want = tuple(h - s + 1 + 2 * gh for s, h, gh in zip(lo, hi, ghost)) + (ncomp,)
                assert a.shape == want, (a.shape, want)
                chunks.append(a.ravel(order="F"))                  # component slowest, Fortran order inside the box
                offs[q + 1] = offs[q] + chunks[-1].size
            data = np.concatenate(chunks) if chunks else np.zeros(0)
            W.dataset(g, "boxes", W.box, np.ascontiguousarray(bx), len(boxes))
            W.dataset(g, "data:datatype=0", W.DBL, np.ascontiguousarray(data))
            W.dataset(g, "data:offsets=0", W.I64, offs)
            W.dataset(g, "Processors", W.INT, np.zeros(len(boxes), dtype=np.int32))
            a = W.group(g, "data_attributes")
            W.attr_int(a, "comps", ncomp)
            W.attr_intvect(a, "ghost", ghost)
            W.attr_intvect(a, "outputGhost", ghost)
            W.attr_str(a, "objectType", "FArrayBox")
            W.L.H5Gclose(a)
            W.L.H5Gclose(g)
    finally:
        W.close()


def read_level(path, l):
    """-> dict(boxes (n, 6) int32, data float64, offsets int64, attrs) of /level_l: what the test reads back"""
    L = lib()
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    assert f >= 0
    INT, DBL, I64 = _t("H5T_NATIVE_INT_g"), _t("H5T_NATIVE_DOUBLE_g"), _t("H5T_NATIVE_LLONG_g")
    g = L.H5Gopen2(f, ("level_%d" % l).encode(), H5P_DEFAULT)
    assert g >= 0

    def dset(name, typ, dtype, width=1):
        d = L.H5Dopen2(g, name.encode(), H5P_DEFAULT)
        assert d >= 0, name
        s = L.H5Dget_space(d)
        n = L.H5Sget_simple_extent_npoints(s)
        out = np.zeros((n, width) if width > 1 else n, dtype=dtype)
        if n:
            assert L.H5Dread(d, typ, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Sclose(s)
        L.H5Dclose(d)
        return out
    box_t = L.H5Tcreate(H5T_COMPOUND, 24)
    for q, n in enumerate(("lo_i", "lo_j", "lo_k", "hi_i", "hi_j", "hi_k")):
        L.H5Tinsert(box_t, n.encode(), 4 * q, INT)
    rv_t = L.H5Tcreate(H5T_COMPOUND, 24)
    for q, n in enumerate(("x", "y", "z")):
        L.H5Tinsert(rv_t, n.encode(), 8 * q, DBL)
    out = {"boxes": dset("boxes", box_t, np.int32, 6), "data": dset("data:datatype=0", DBL, np.float64),
           "offsets": dset("data:offsets=0", I64, np.int64)}
    dxv = (C.c_double * 3)()
    a = L.H5Aopen(g, b"vec_dx", H5P_DEFAULT)
    assert a >= 0 and L.H5Aread(a, rv_t, C.cast(dxv, C.c_void_p)) >= 0
    L.H5Aclose(a)
    dom = (C.c_int * 6)()
    a = L.H5Aopen(g, b"prob_domain", H5P_DEFAULT)
    assert a >= 0 and L.H5Aread(a, box_t, C.cast(dom, C.c_void_p)) >= 0
    L.H5Aclose(a)
    out["vec_dx"], out["prob_domain"] = tuple(dxv), tuple(dom)
    L.H5Tclose(box_t)
    L.H5Tclose(rv_t)
    L.H5Gclose(g)
    L.H5Fclose(f)
    return out
