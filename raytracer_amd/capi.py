"""ctypes binding of librt_dp.so (the C ABI declared in include/rt_dp.h).

There is no CPU fallback: loading fails loudly when the HIP library has not been
built, and every call goes through the C ABI.

Arrays may be numpy arrays (host pointers: the library stages them) or torch
CUDA tensors (device pointers, RT_DP_F_DEVICE_PTRS: the caller's HBM is used in
place).  A batch must not mix the two kinds.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_DP_LIB") or os.path.join(_HERE, "lib", "librt_dp.so")   # RT_DP_LIB: another build of the same library (A/B runs)

OP_TRIANGLE, OP_QUADBOX, OP_EUCLIDEAN, OP_ANGULAR = 0, 1, 2, 3
F_DEVICE_PTRS = 0x1
F_DERIVE_RAY = 0x2
F_BOX_TILED = 0x4
F_TRI_TILED = 0x8
TILE_OPS = 8192
BOX_TILE_IN_WORDS, BOX_TILE_OUT_WORDS, TRI_TILE_IN_WORDS, TRI_TILE_OUT_WORDS = 34, 5, 16, 3


def tiled_words(n, words):
    """RT_DP_TILED_WORDS: 32-bit words of a tiled buffer of n ops with `words` words per op."""
    return ((int(n) + TILE_OPS - 1) // TILE_OPS) * TILE_OPS * int(words)

OK, EINVAL, ENULL, EOPCODE, EHIP, ENODEV, EALIGN, ENOMEM = 0, -1, -2, -3, -4, -5, -6, -7

# operand order of the SoA segments (matches rt_dp_box_in / rt_dp_tri_in)
BOX_FIELDS = (["org_x", "org_y", "org_z", "dir_x", "dir_y", "dir_z", "inv_x", "inv_y", "inv_z", "extent"] +
              ["b%d_%s" % (b, f) for b in range(4) for f in ("x_min", "y_min", "z_min", "x_max", "y_max", "z_max")])
TRI_FIELDS = ["org_x", "org_y", "org_z", "shear_x", "shear_y", "shear_z",
              "ax", "ay", "az", "bx", "by", "bz", "cx", "cy", "cz"]

_fp = C.c_void_p


class BoxIn(C.Structure):
    _fields_ = [("org", _fp * 3), ("dir", _fp * 3), ("inv", _fp * 3), ("extent", _fp), ("aabb", (_fp * 6) * 4)]


class BoxOut(C.Structure):
    _fields_ = [("tmin_out", _fp * 4), ("flags", _fp)]


class TriIn(C.Structure):
    _fields_ = [("org", _fp * 3), ("shear", _fp * 3), ("vtx", (_fp * 3) * 3), ("k", _fp), ("dir", _fp * 3)]


class TriOut(C.Structure):
    _fields_ = [("t_num", _fp), ("t_denom", _fp), ("hit", _fp)]


class VecIn(C.Structure):
    _fields_ = [("a", _fp), ("b", _fp), ("ctl", _fp)]


class EucOut(C.Structure):
    _fields_ = [("accumulator", _fp)]


class AngOut(C.Structure):
    _fields_ = [("dot_product", _fp), ("norm", _fp)]


class OpBatch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("n", C.c_uint64 * 4),
                ("tri", TriIn), ("box", BoxIn), ("euc", VecIn), ("ang", VecIn)]


class ResultBatch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("tri", TriOut), ("box", BoxOut), ("euc", EucOut), ("ang", AngOut)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("io_recorded_float", C.c_uint32),
                ("internal_recorded_float", C.c_uint32), ("support_euclidean", C.c_uint32), ("reserved", C.c_uint32)]


class RayPrep(C.Structure):
    _fields_ = [("dir", _fp * 3), ("inv", _fp * 3), ("k", _fp), ("shear", _fp * 3), ("extent", _fp)]


class F3(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class Ray(C.Structure):
    _fields_ = [("origin", F3), ("dir", F3), ("inv", F3), ("extent", C.c_uint32),
                ("kx", C.c_uint32), ("ky", C.c_uint32), ("kz", C.c_uint32), ("shear", F3)]


class AABB(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("x_min", "y_min", "z_min", "x_max", "y_max", "z_max")]


class Triangle(C.Structure):
    _fields_ = [("A", F3), ("B", F3), ("C", F3)]


class InputBundle(C.Structure):
    _fields_ = [("opcode", C.c_uint32), ("ray", Ray), ("aabb", AABB * 4), ("triangle", Triangle),
                ("euclidean_a", C.c_uint32 * 16), ("euclidean_b", C.c_uint32 * 16),
                ("euclidean_mask", C.c_uint32), ("euclidean_reset_accum", C.c_uint32)]


class OutputBundle(C.Structure):
    _fields_ = [("opcode", C.c_uint32), ("tmin_out", C.c_uint32 * 4), ("isIntersect", C.c_uint32 * 4),
                ("boxIndex", C.c_uint32 * 4), ("t_num", C.c_uint32), ("t_denom", C.c_uint32),
                ("triangle_hit", C.c_uint32), ("euclidean_accumulator", C.c_uint32),
                ("euclidean_reset_accum", C.c_uint32), ("angular_dot_product", C.c_uint32),
                ("angular_norm", C.c_uint32), ("angular_reset_accum", C.c_uint32)]


assert C.sizeof(InputBundle) == 84 * 4 and C.sizeof(OutputBundle) == 21 * 4


TRACE_WAVEFRONT = 1


class Bvh4(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_nodes", C.c_uint32), ("n_tris", C.c_uint32), ("reserved", C.c_uint32),
                ("nodes", C.c_void_p), ("tris", C.c_void_p)]


class KnnTree(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_leaves", C.c_uint32), ("leaf_size", C.c_uint32), ("reserved", C.c_uint32),
                ("points", C.c_void_p), ("index", C.c_void_p), ("leaf_lo", C.c_void_p), ("leaf_hi", C.c_void_p)]

EXPORTS = ["rt_dp_create", "rt_dp_destroy", "rt_dp_eval", "rt_dp_eval_ctx", "rt_dp_launch", "rt_dp_synchronize", "rt_dp_tile_pack", "rt_dp_tile_unpack",
           "rt_dp_reset", "rt_dp_get_accum", "rt_dp_set_accum", "rt_dp_eval_bundles", "rt_dp_set_tuning", "rt_dp_set_option", "rt_dp_get_option", "rt_dp_knn_euclid16", "rt_dp_knn_angular16", "rt_dp_knn_tree_euclid16", "rt_dp_knn_tree_grouped_euclid16", "rt_dp_trace_closest", "rt_dp_box_compact", "rt_dp_box_compact_tiled", "rt_dp_box_compact_static", "rt_dp_beat_offsets_host", "rt_dp_pairs_to_beats", "rt_dp_beats_to_pairs", "rt_dp_bvh4_build_device", "rt_dp_knn_tree_build_device", "rt_dp_knn_home_leaf", "rt_dp_ray_prepare", "rt_dp_eval_bundles_device",
           "rt_dp_knn_tree_leaves", "rt_dp_knn_tree_build_host", "rt_dp_knn_home_leaf_host", "rt_dp_bvh4_max_nodes", "rt_dp_bvh4_build_host",
           "rt_dp_shard_plan", "rt_dp_shard_plan_records", "rt_dp_kernel_name", "rt_dp_knn_tree_angular16",
           "rt_dp_last_error", "rt_dp_version", "rt_dp_device_count"]


class ShardRange(C.Structure):
    _fields_ = [("start", C.c_uint64 * 4), ("stop", C.c_uint64 * 4)]


class RtDpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt_dp error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load librt_dp.so.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("librt_dp.so not found at %s: build it with `python -m raytracer_amd.build` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_dp_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.rt_dp_destroy.argtypes = [vp]
    L.rt_dp_destroy.restype = None
    L.rt_dp_eval.argtypes = [C.POINTER(OpBatch), C.POINTER(ResultBatch)]
    L.rt_dp_eval_ctx.argtypes = [vp, C.POINTER(OpBatch), C.POINTER(ResultBatch)]
    L.rt_dp_launch.argtypes = [vp, C.POINTER(OpBatch), C.POINTER(ResultBatch), vp]
    L.rt_dp_synchronize.argtypes = [vp]
    L.rt_dp_tile_pack.argtypes = [vp, C.POINTER(vp), C.c_uint32, C.c_uint64, vp, vp]
    L.rt_dp_tile_unpack.argtypes = [vp, vp, C.POINTER(vp), C.c_uint32, C.c_uint64, vp]
    L.rt_dp_reset.argtypes = [vp]
    L.rt_dp_get_accum.argtypes = [vp, C.POINTER(C.c_float * 3)]
    L.rt_dp_set_accum.argtypes = [vp, C.POINTER(C.c_float * 3)]
    L.rt_dp_eval_bundles.argtypes = [vp, vp, vp, C.c_uint64]
    L.rt_dp_eval_bundles_device.argtypes = [vp, vp, vp, C.c_uint64]
    L.rt_dp_set_tuning.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.rt_dp_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.rt_dp_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.rt_dp_ray_prepare.argtypes = [vp, C.POINTER(RayPrep), C.c_uint64, vp]
    L.rt_dp_knn_euclid16.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rt_dp_knn_angular16.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rt_dp_knn_tree_euclid16.argtypes = [vp, C.POINTER(KnnTree), vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.rt_dp_knn_tree_angular16.argtypes = [vp, C.POINTER(KnnTree), vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    L.rt_dp_knn_tree_grouped_euclid16.argtypes = [vp, C.POINTER(KnnTree), vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(C.c_uint64 * 4)]
    L.rt_dp_bvh4_build_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_uint32), vp, vp]
    L.rt_dp_beat_offsets_host.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.rt_dp_beat_offsets_host.restype = C.c_uint64
    L.rt_dp_pairs_to_beats.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp, vp]
    L.rt_dp_beats_to_pairs.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
    L.rt_dp_knn_tree_build_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.rt_dp_knn_home_leaf.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.rt_dp_box_compact.argtypes = [vp, C.POINTER(BoxOut), C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rt_dp_box_compact_tiled.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rt_dp_box_compact_static.argtypes = [vp, C.POINTER(BoxOut), C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.rt_dp_trace_closest.argtypes = [vp, C.POINTER(Bvh4), C.POINTER(vp * 3), C.POINTER(vp * 3), C.c_uint32, C.c_float, C.c_uint32, vp, vp,
                                      C.POINTER(C.c_uint64 * 4)]
    L.rt_dp_knn_tree_leaves.argtypes = [C.c_uint32, C.c_uint32]
    L.rt_dp_knn_tree_build_host.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.rt_dp_knn_home_leaf_host.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rt_dp_bvh4_max_nodes.argtypes = [C.c_uint32]
    L.rt_dp_bvh4_build_host.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.rt_dp_kernel_name.argtypes = [vp, C.POINTER(OpBatch), C.POINTER(ResultBatch), C.c_int, C.c_char_p, C.c_uint64]
    L.rt_dp_shard_plan.argtypes = [C.POINTER(C.c_uint64 * 4), vp, vp, C.c_uint32, C.c_uint32, C.POINTER(ShardRange)]
    L.rt_dp_shard_plan_records.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.rt_dp_last_error.restype = C.c_char_p
    L.rt_dp_version.restype = C.c_char_p
    for name in EXPORTS:
        if name in ("rt_dp_knn_tree_leaves", "rt_dp_bvh4_max_nodes"):
            getattr(L, name).restype = C.c_uint32
        elif name not in ("rt_dp_destroy", "rt_dp_last_error", "rt_dp_version"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise RtDpError(rc, load().rt_dp_last_error().decode())


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, kinds):
    """Pointer of a numpy array (host) or torch tensor (device); records which in kinds."""
    if x is None:
        return None
    if _is_torch(x):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        kinds.add("cuda" if x.is_cuda else "host")
        return x.data_ptr()
    if not isinstance(x, np.ndarray) or not x.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be a C-contiguous numpy array or a contiguous torch tensor")
    kinds.add("host")
    return x.ctypes.data


def _words(x):
    """Number of elements of a buffer of 32-bit words (torch tensor or numpy array)."""
    if _is_torch(x):
        assert x.element_size() == 4
        return x.numel()
    assert x.itemsize == 4
    return x.size


class Batch:
    """Builder for one rt_dp_op_batch / rt_dp_result_batch pair.  Keeps every array alive."""

    def __init__(self):
        self.ob = OpBatch()
        self.rb = ResultBatch()
        self.ob.struct_size = C.sizeof(OpBatch)
        self.rb.struct_size = C.sizeof(ResultBatch)
        self._keep = []
        self._kinds = set()
        self._derive = False
        self._tiled = 0

    def _p(self, x):
        self._keep.append(x)
        return _ptr(x, self._kinds)

    def set_box(self, fields, tmin, flags, n=None):
        """fields: 34 arrays in BOX_FIELDS order; tmin: 4 output arrays; flags: uint32 output."""
        assert len(fields) == 34 and len(tmin) == 4
        b = self.ob.box
        for a in range(3):
            b.org[a] = self._p(fields[a])
            b.dir[a] = self._p(fields[3 + a])
            b.inv[a] = None if self._derive else self._p(fields[6 + a])
        b.extent = self._p(fields[9])
        for bi in range(4):
            for f in range(6):
                b.aabb[bi][f] = self._p(fields[10 + 6 * bi + f])
        for k in range(4):
            self.rb.box.tmin_out[k] = self._p(tmin[k])
        self.rb.box.flags = self._p(flags)
        self.ob.n[OP_QUADBOX] = int(fields[0].shape[0]) if n is None else int(n)

    def set_tri(self, fields, k, t_num, t_denom, hit, n=None):
        """fields: 15 arrays in TRI_FIELDS order; k: packed kx|ky<<2|kz<<4."""
        assert len(fields) == 15
        t = self.ob.tri
        for a in range(3):
            t.org[a] = self._p(fields[a])
            if self._derive:
                t.dir[a] = self._p(fields[3 + a])
            else:
                t.shear[a] = self._p(fields[3 + a])
        for v in range(3):
            for a in range(3):
                t.vtx[v][a] = self._p(fields[6 + 3 * v + a])
        t.k = None if self._derive else self._p(k)
        self.rb.tri.t_num = self._p(t_num)
        self.rb.tri.t_denom = self._p(t_denom)
        self.rb.tri.hit = self._p(hit)
        self.ob.n[OP_TRIANGLE] = int(fields[0].shape[0]) if n is None else int(n)

    def set_box_tiled(self, operands, results, n):
        """RT_DP_F_BOX_TILED: operands / results are buffers (CUDA tensors or numpy arrays) of tiled_words(n, 34) /
        tiled_words(n, 5) 32-bit words."""
        assert _words(operands) >= tiled_words(n, BOX_TILE_IN_WORDS) and _words(results) >= tiled_words(n, BOX_TILE_OUT_WORDS)
        self.ob.box.org[0] = self._p(operands)
        self.rb.box.tmin_out[0] = self._p(results)
        self.ob.n[OP_QUADBOX] = int(n)
        self._tiled |= F_BOX_TILED

    def set_tri_tiled(self, operands, results, n):
        """RT_DP_F_TRI_TILED: operands / results are buffers (CUDA tensors or numpy arrays) of tiled_words(n, 16) /
        tiled_words(n, 3) 32-bit words."""
        assert _words(operands) >= tiled_words(n, TRI_TILE_IN_WORDS) and _words(results) >= tiled_words(n, TRI_TILE_OUT_WORDS)
        self.ob.tri.org[0] = self._p(operands)
        self.rb.tri.t_num = self._p(results)
        self.ob.n[OP_TRIANGLE] = int(n)
        self._tiled |= F_TRI_TILED

    def set_euclid(self, a, b, ctl, acc, n=None):
        self.ob.euc.a, self.ob.euc.b, self.ob.euc.ctl = self._p(a), self._p(b), self._p(ctl)
        self.rb.euc.accumulator = self._p(acc)
        self.ob.n[OP_EUCLIDEAN] = int(ctl.shape[0]) if n is None else int(n)

    def set_angular(self, q, c, ctl, dot, norm, n=None):
        self.ob.ang.a, self.ob.ang.b, self.ob.ang.ctl = self._p(q), self._p(c), self._p(ctl)
        self.rb.ang.dot_product = self._p(dot)
        self.rb.ang.norm = self._p(norm)
        self.ob.n[OP_ANGULAR] = int(ctl.shape[0]) if n is None else int(n)

    def finalize(self):
        if len(self._kinds) > 1:
            raise ValueError("a batch must not mix host and device arrays")
        self.ob.flags = (F_DEVICE_PTRS if "cuda" in self._kinds else 0) | (F_DERIVE_RAY if self._derive else 0) | self._tiled
        return self

    def derive_ray(self, on=True):
        """RT_DP_F_DERIVE_RAY: inv / shear / k come from dir inside the kernels.  In this mode set_box ignores
        fields[6:9] (they may be None) and set_tri takes the ray direction in fields[3:6], k may be None."""
        self._derive = bool(on)
        return self


class Context:
    """One datapath instance (rt_dp_ctx): owns the accumulator registers."""

    def __init__(self, device=-1, support_euclidean=16, io_recorded_float=0, internal_recorded_float=1):
        L = load()
        cfg = Config(C.sizeof(Config), device, io_recorded_float, internal_recorded_float, support_euclidean, 0)
        h = C.c_void_p()
        _check(L.rt_dp_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_dp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def eval(self, batch):
        batch.finalize()
        _check(self._L.rt_dp_eval_ctx(self._h, C.byref(batch.ob), C.byref(batch.rb)))

    def launch(self, batch, stream=None):
        """Asynchronous launch on `stream` (int hipStream_t handle, e.g. torch's cuda_stream)."""
        batch.finalize()
        _check(self._L.rt_dp_launch(self._h, C.byref(batch.ob), C.byref(batch.rb), C.c_void_p(stream) if stream else None))

    def synchronize(self):
        _check(self._L.rt_dp_synchronize(self._h))

    def tile_pack(self, arrays, tiled, n=None, stream=None):
        """arrays: CUDA tensors of n 32-bit elements each; tiled: CUDA tensor of tiled_words(n, len(arrays)) words."""
        n = int(arrays[0].shape[0]) if n is None else int(n)
        keep = set()
        ptrs = (C.c_void_p * len(arrays))(*[_ptr(a, keep) for a in arrays])
        tp = _ptr(tiled, keep)
        if keep != {"cuda"}:
            raise ValueError("tile_pack needs CUDA tensors")
        assert all(a.element_size() == 4 and a.numel() >= n for a in arrays) and tiled.element_size() == 4
        assert tiled.numel() >= tiled_words(n, len(arrays))
        _check(self._L.rt_dp_tile_pack(self._h, ptrs, len(arrays), n, tp, C.c_void_p(stream) if stream else None))

    def tile_unpack(self, tiled, arrays, n=None, stream=None):
        n = int(arrays[0].shape[0]) if n is None else int(n)
        keep = set()
        ptrs = (C.c_void_p * len(arrays))(*[_ptr(a, keep) for a in arrays])
        tp = _ptr(tiled, keep)
        if keep != {"cuda"}:
            raise ValueError("tile_unpack needs CUDA tensors")
        assert all(a.element_size() == 4 and a.numel() >= n for a in arrays) and tiled.element_size() == 4
        assert tiled.numel() >= tiled_words(n, len(arrays))
        _check(self._L.rt_dp_tile_unpack(self._h, tp, ptrs, len(arrays), n, C.c_void_p(stream) if stream else None))

    def reset(self):
        _check(self._L.rt_dp_reset(self._h))

    def get_accum(self):
        a = (C.c_float * 3)()
        _check(self._L.rt_dp_get_accum(self._h, C.byref(a)))
        return np.array(list(a), np.float32)

    def set_accum(self, acc):
        a = (C.c_float * 3)(*[float(x) for x in acc])
        _check(self._L.rt_dp_set_accum(self._h, C.byref(a)))

    def set_tuning(self, ops_per_lane=0, nontemporal=0, vec_tile=0, variant=0, vec_passes=0):
        _check(self._L.rt_dp_set_tuning(self._h, ops_per_lane, nontemporal, vec_tile))
        _check(self._L.rt_dp_set_option(self._h, b"variant", variant))
        _check(self._L.rt_dp_set_option(self._h, b"vec_passes", vec_passes))

    def set_option(self, key, value):
        _check(self._L.rt_dp_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        _check(self._L.rt_dp_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def kernel_name(self, batch, opcode):
        """Name (as rocprofv3 prints it) of the kernel a launch of `batch` runs for the segment of `opcode` (0..3)."""
        batch.finalize()
        buf = C.create_string_buffer(128)
        _check(self._L.rt_dp_kernel_name(self._h, C.byref(batch.ob), C.byref(batch.rb), int(opcode), buf, 128))
        return buf.value.decode()

    def eval_bundles_device(self, d_in, d_out):
        """Record stream resident in HBM: d_in int32 CUDA tensor [n, 84], d_out int32 CUDA tensor [n, 21]."""
        assert d_in.is_cuda and d_out.is_cuda and d_in.is_contiguous() and d_out.is_contiguous()
        assert d_in.shape[1] == 84 and tuple(d_out.shape) == (d_in.shape[0], 21)
        _check(self._L.rt_dp_eval_bundles_device(self._h, d_in.data_ptr(), d_out.data_ptr(), int(d_in.shape[0])))

    def ray_prepare(self, dirs, inv=None, k=None, shear=None, extent=None, stream=None):
        """SW_Ray derivation on the GPU.  dirs / inv / shear: lists of three float32 CUDA tensors; k int32; extent float32."""
        rp = RayPrep()
        keep = set()
        for c in range(3):
            rp.dir[c] = _ptr(dirs[c], keep)
            if inv is not None:
                rp.inv[c] = _ptr(inv[c], keep)
            if shear is not None:
                rp.shear[c] = _ptr(shear[c], keep)
        rp.k = _ptr(k, keep)
        rp.extent = _ptr(extent, keep)
        if keep != {"cuda"}:
            raise ValueError("ray_prepare needs CUDA tensors")
        _check(self._L.rt_dp_ray_prepare(self._h, C.byref(rp), int(dirs[0].shape[0]), C.c_void_p(stream) if stream else None))

    def knn_angular16(self, points, queries, out_neg_score, out_idx, index_base=0, splits=1, stream=None):
        """Top-16 points per query by the angular op's dot / sqrt(norm) (largest first); out_neg_score holds minus the score."""
        for t in (points, queries, out_neg_score, out_idx):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("knn_angular16 needs contiguous CUDA tensors")
        n, m = int(points.shape[0]), int(queries.shape[0])
        assert tuple(out_neg_score.shape) == (splits, m, 16) and tuple(out_idx.shape) == (splits, m, 16)
        _check(self._L.rt_dp_knn_angular16(self._h, points.data_ptr(), n, queries.data_ptr(), m, index_base, splits,
                                           out_neg_score.data_ptr(), out_idx.data_ptr(), C.c_void_p(stream) if stream else None))

    def knn_euclid16(self, points, queries, out_dist, out_idx, index_base=0, splits=1, stream=None):
        """Top-16 nearest points per query (torch CUDA tensors): points [n,16], queries [m,16] float32,
        out_dist [splits,m,16] float32, out_idx [splits,m,16] int32 (bit image of uint32).  Asynchronous."""
        for t in (points, queries, out_dist, out_idx):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("knn_euclid16 needs contiguous CUDA tensors")
        n, m = int(points.shape[0]), int(queries.shape[0])
        assert tuple(out_dist.shape) == (splits, m, 16) and tuple(out_idx.shape) == (splits, m, 16)
        _check(self._L.rt_dp_knn_euclid16(self._h, points.data_ptr(), n, queries.data_ptr(), m, index_base, splits,
                                          out_dist.data_ptr(), out_idx.data_ptr(), C.c_void_p(stream) if stream else None))

    def knn_tree_euclid16(self, tree, queries, home_leaf, bound, out_dist, out_idx, splits=1, stream=None):
        """Tree-pruned top-16 (torch CUDA tensors).  tree: raytracer_amd.knn_tree.KnnTree; queries [m,16] float32 sorted by
        home_leaf [m] int32; bound [m] int32 scratch; out_dist / out_idx [splits,m,16].  Asynchronous."""
        for t in (queries, home_leaf, bound, out_dist, out_idx, tree.points, tree.index, tree.leaf_lo, tree.leaf_hi):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("knn_tree_euclid16 needs contiguous CUDA tensors")
        m = int(queries.shape[0])
        assert tuple(out_dist.shape) == (splits, m, 16) and tuple(out_idx.shape) == (splits, m, 16)
        assert home_leaf.shape[0] == m and bound.shape[0] == m
        ct = KnnTree(C.sizeof(KnnTree), tree.n_leaves, tree.leaf_size, 0, tree.points.data_ptr(), tree.index.data_ptr(),
                     tree.leaf_lo.data_ptr(), tree.leaf_hi.data_ptr())
        _check(self._L.rt_dp_knn_tree_euclid16(self._h, C.byref(ct), queries.data_ptr(), home_leaf.data_ptr(), m, splits, bound.data_ptr(),
                                               out_dist.data_ptr(), out_idx.data_ptr(), C.c_void_p(stream) if stream else None))

    def knn_tree_angular16(self, tree, queries, home_leaf, bound, out_neg_score, out_idx, splits=1, leaves_scanned=None, stream=None):
        """rt_dp_knn_tree_angular16: tree is a raytracer_amd.knn_tree.KnnTree; out_* [splits, m, 16]; leaves_scanned: int64 CUDA tensor [1] or None."""
        t = KnnTree(C.sizeof(KnnTree), tree.n_leaves, tree.leaf_size, 0, tree.points.data_ptr(), tree.index.data_ptr(), tree.leaf_lo.data_ptr(),
                    tree.leaf_hi.data_ptr())
        m = int(queries.shape[0])
        assert out_neg_score.numel() == splits * m * 16 and out_idx.numel() == splits * m * 16 and bound.numel() >= m and home_leaf.numel() == m
        _check(self._L.rt_dp_knn_tree_angular16(self._h, C.byref(t), queries.data_ptr(), home_leaf.data_ptr(), m, splits, bound.data_ptr(),
                                                out_neg_score.data_ptr(), out_idx.data_ptr(), None if leaves_scanned is None else leaves_scanned.data_ptr(),
                                                stream))

    def knn_tree_grouped_euclid16(self, tree, queries, home_leaf, out_dist, out_idx, walk_leaves=16):
        """Leaf-major exact top-16 (torch CUDA tensors): queries [m,16] sorted by home_leaf [m]; out_dist / out_idx [m,16].
        Returns {pairs, groups, candidate_slots, rounds}.  Synchronous."""
        for t in (queries, home_leaf, out_dist, out_idx, tree.points, tree.index, tree.leaf_lo, tree.leaf_hi):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("knn_tree_grouped_euclid16 needs contiguous CUDA tensors")
        m = int(queries.shape[0])
        assert tuple(out_dist.shape) == (m, 16) and tuple(out_idx.shape) == (m, 16) and home_leaf.shape[0] == m
        ct = KnnTree(C.sizeof(KnnTree), tree.n_leaves, tree.leaf_size, 0, tree.points.data_ptr(), tree.index.data_ptr(),
                     tree.leaf_lo.data_ptr(), tree.leaf_hi.data_ptr())
        stats = (C.c_uint64 * 4)()
        _check(self._L.rt_dp_knn_tree_grouped_euclid16(self._h, C.byref(ct), queries.data_ptr(), home_leaf.data_ptr(), m, int(walk_leaves),
                                                       out_dist.data_ptr(), out_idx.data_ptr(), C.byref(stats)))
        return {"pairs": stats[0], "groups": stats[1], "candidate_slots": stats[2], "rounds": stats[3]}

    def box_compact(self, tmin, flags, out_item, out_child, out_tmin=None, item=None, children=None):
        """Dense (item, child[, tmin]) list of the intersected children of a Quadbox batch (torch CUDA tensors).
        tmin: the four result arrays (or None when out_tmin is None); flags: the packed result word.  Returns the
        number of entries the batch has (only out_item.shape[0] of them are written)."""
        ro = BoxOut()
        keep = set()
        if tmin is not None:
            for k in range(4):
                ro.tmin_out[k] = _ptr(tmin[k], keep)
        ro.flags = _ptr(flags, keep)
        n = int(flags.shape[0])
        count = C.c_uint64(0)
        p = lambda t: None if t is None else t.data_ptr()
        _check(self._L.rt_dp_box_compact(self._h, C.byref(ro), n, p(item), p(children), p(out_item), p(out_child), p(out_tmin),
                                         int(out_item.shape[0]), C.byref(count)))
        return int(count.value)

    def box_compact_static(self, tmin, flags, out_item, out_child, d_count, out_tmin=None, item=None, children=None, n=None, stream=None):
        """rt_dp_box_compact_static: as box_compact, but allocation-free and asynchronous; d_count is an int64 CUDA
        tensor of one element that receives the number of entries.  n defaults to flags.shape[0]."""
        ro = BoxOut()
        keep = set()
        if tmin is not None:
            for k in range(4):
                ro.tmin_out[k] = _ptr(tmin[k], keep)
        ro.flags = _ptr(flags, keep)
        p = lambda t: None if t is None else t.data_ptr()
        _check(self._L.rt_dp_box_compact_static(self._h, C.byref(ro), int(flags.shape[0]) if n is None else int(n), p(item), p(children),
                                                p(out_item), p(out_child), p(out_tmin), int(out_item.shape[0]), p(d_count),
                                                C.c_void_p(stream) if stream else None))

    def box_compact_tiled(self, results, n, out_item, out_child, out_tmin=None, item=None, children=None):
        """box_compact over the tiled result buffer of a RT_DP_F_BOX_TILED segment (CUDA tensor of tiled_words(n, 5) words)."""
        assert results.is_cuda and results.is_contiguous() and results.element_size() == 4 and results.numel() >= tiled_words(n, BOX_TILE_OUT_WORDS)
        count = C.c_uint64(0)
        p = lambda t: None if t is None else t.data_ptr()
        _check(self._L.rt_dp_box_compact_tiled(self._h, results.data_ptr(), int(n), p(item), p(children), p(out_item), p(out_child), p(out_tmin),
                                               int(out_item.shape[0]), C.byref(count)))
        return int(count.value)

    def pairs_to_beats(self, a, b, pair_offsets, width):
        """Vector pairs (flat float32 CUDA tensors a, b; pair_offsets: int64 numpy array [n_pairs + 1]) split into
        beats of `width` elements.  Returns (a_beats [n, width], b_beats [n, width], ctl int32 [n], beat_offsets int64
        CUDA tensor [n_pairs + 1])."""
        import torch
        po = np.ascontiguousarray(pair_offsets, np.uint64)
        n_pairs = po.shape[0] - 1
        bo = np.zeros(n_pairs + 1, np.uint64)
        n_beats = int(self._L.rt_dp_beat_offsets_host(po.ctypes.data, n_pairs, width, bo.ctypes.data))
        d_po = torch.from_numpy(po.view(np.int64)).to(a.device)
        d_bo = torch.from_numpy(bo.view(np.int64)).to(a.device)
        oa = torch.empty((n_beats, width), dtype=torch.float32, device=a.device)
        ob = torch.empty((n_beats, width), dtype=torch.float32, device=a.device)
        ctl = torch.empty(n_beats, dtype=torch.int32, device=a.device)
        torch.cuda.synchronize()
        _check(self._L.rt_dp_pairs_to_beats(self._h, a.data_ptr(), b.data_ptr(), d_po.data_ptr(), d_bo.data_ptr(), n_pairs, n_beats, width,
                                            oa.data_ptr(), ob.data_ptr(), ctl.data_ptr(), None))
        self.synchronize()
        return oa, ob, ctl, d_bo

    def beats_to_pairs(self, values, beat_offsets):
        """Per pair, the value of its last beat (float32 CUDA tensor [n_pairs])."""
        import torch
        n_pairs = int(beat_offsets.shape[0]) - 1
        out = torch.empty(n_pairs, dtype=torch.float32, device=values.device)
        torch.cuda.synchronize()
        _check(self._L.rt_dp_beats_to_pairs(self._h, values.data_ptr(), beat_offsets.data_ptr(), n_pairs, out.data_ptr(), None))
        self.synchronize()
        return out

    def knn_tree_build_device(self, points, leaf_size=128, index_base=0):
        """k-d index over points (float32 CUDA tensor [n, 16]) built on the GPU.  Returns a dict of CUDA tensors:
        points [L * leaf_size, 16], index int32, leaf_lo / leaf_hi [L, 16], split_dim int32 / split_val [L - 1] (heap
        order), and n_leaves."""
        import torch
        assert points.is_cuda and points.is_contiguous() and points.dtype == torch.float32 and points.shape[1] == 16
        n = int(points.shape[0])
        leaves = int(self._L.rt_dp_knn_tree_leaves(n, leaf_size))
        dev = points.device
        t = {"n_leaves": leaves, "leaf_size": leaf_size,
             "points": torch.empty((leaves * leaf_size, 16), dtype=torch.float32, device=dev),
             "index": torch.empty(leaves * leaf_size, dtype=torch.int32, device=dev),
             "leaf_lo": torch.empty((leaves, 16), dtype=torch.float32, device=dev),
             "leaf_hi": torch.empty((leaves, 16), dtype=torch.float32, device=dev),
             "split_dim": torch.zeros(max(1, leaves - 1), dtype=torch.int32, device=dev),
             "split_val": torch.zeros(max(1, leaves - 1), dtype=torch.float32, device=dev)}
        _check(self._L.rt_dp_knn_tree_build_device(self._h, points.data_ptr(), n, leaf_size, index_base, t["points"].data_ptr(), t["index"].data_ptr(),
                                                   t["leaf_lo"].data_ptr(), t["leaf_hi"].data_ptr(), t["split_dim"].data_ptr(), t["split_val"].data_ptr()))
        return t

    def knn_home_leaf(self, queries, tree):
        """The leaf every query (float32 CUDA tensor [m, 16]) reaches in a tree from knn_tree_build_device (int32 [m])."""
        import torch
        m = int(queries.shape[0])
        out = torch.empty(m, dtype=torch.int32, device=queries.device)
        torch.cuda.synchronize()
        _check(self._L.rt_dp_knn_home_leaf(self._h, queries.data_ptr(), m, tree["n_leaves"], tree["split_dim"].data_ptr(), tree["split_val"].data_ptr(),
                                           out.data_ptr(), None))
        self.synchronize()
        return out

    def bvh4_build_device(self, tris, leaf_size=4):
        """Hierarchy over tris (float32 CUDA tensor [n, 9]) built on the GPU.  Returns (nodes [m, 32] float32,
        tris12 [n, 12] float32 in leaf order, order [n] int32: the caller's index of each stored triangle)."""
        import torch
        assert tris.is_cuda and tris.is_contiguous() and tris.dtype == torch.float32 and tris.shape[1] == 9
        n = int(tris.shape[0])
        if not (1 <= n <= 1 << 28 and 1 <= leaf_size <= 8):
            raise ValueError("n must be 1 .. 2^28 and leaf_size 1 .. 8")
        cap = int(self._L.rt_dp_bvh4_max_nodes(n))
        nodes = torch.empty((cap, 32), dtype=torch.float32, device=tris.device)
        tris12 = torch.empty((n, 12), dtype=torch.float32, device=tris.device)
        order = torch.empty(n, dtype=torch.int32, device=tris.device)
        m = C.c_uint32(0)
        _check(self._L.rt_dp_bvh4_build_device(self._h, tris.data_ptr(), n, leaf_size, nodes.data_ptr(), C.byref(m), tris12.data_ptr(), order.data_ptr()))
        nodes = nodes[:m.value]
        return nodes, tris12, order

    def trace_closest(self, nodes, tris, origin, direction, extent, out_t, out_tri, wavefront=False):
        """Closest hit of every ray in a 4-wide hierarchy (torch CUDA tensors).  nodes [m,32] float32, tris [n,12] float32,
        origin / direction: lists of three float32 tensors [r]; out_t float32 [r], out_tri int32 [r] (bit image of uint32).
        Returns {levels, box_ops, tri_ops, widest_queue}.  Synchronous."""
        for t in [nodes, tris, out_t, out_tri] + list(origin) + list(direction):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("trace_closest needs contiguous CUDA tensors")
        r = int(origin[0].shape[0])
        assert nodes.shape[1] == 32 and (tris.shape[0] == 0 or tris.shape[1] == 12) and out_t.shape[0] == r and out_tri.shape[0] == r
        bvh = Bvh4(C.sizeof(Bvh4), int(nodes.shape[0]), int(tris.shape[0]), 0, nodes.data_ptr(), tris.data_ptr() if tris.shape[0] else None)
        o = (C.c_void_p * 3)(*[t.data_ptr() for t in origin])
        d = (C.c_void_p * 3)(*[t.data_ptr() for t in direction])
        stats = (C.c_uint64 * 4)()
        _check(self._L.rt_dp_trace_closest(self._h, C.byref(bvh), C.byref(o), C.byref(d), r, float(extent), TRACE_WAVEFRONT if wavefront else 0, out_t.data_ptr(), out_tri.data_ptr(),
                                           C.byref(stats)))
        return {"levels": stats[0], "box_ops": stats[1], "tri_ops": stats[2], "widest_queue": stats[3]}

    def eval_bundles(self, in_bundles):
        """in_bundles: numpy structured array whose itemsize equals sizeof(rt_dp_input_bundle)
        (84 uint32 words per record).  Returns a uint32 array [n, 21] of output bundles."""
        raw = np.ascontiguousarray(in_bundles).view(np.uint32).reshape(-1, 84)
        out = np.zeros((raw.shape[0], 21), np.uint32)
        _check(self._L.rt_dp_eval_bundles(self._h, raw.ctypes.data, out.ctypes.data, raw.shape[0]))
        return out


def version():
    return load().rt_dp_version().decode()


def device_count():
    return load().rt_dp_device_count()


def shard_plan(n_tri, n_box, euc_ctl, ang_ctl, world, box_tiled=False, tri_tiled=False):
    """rt_dp_shard_plan on host arrays (numpy uint32 control words, or None for an empty segment).  Returns, per part,
    {"tri": (start, stop), "box": ..., "euclid": ..., "angular": ...} like raytracer_amd.shard.plan."""
    L = load()
    e = None if euc_ctl is None else np.ascontiguousarray(euc_ctl).view(np.uint32)
    a = None if ang_ctl is None else np.ascontiguousarray(ang_ctl).view(np.uint32)
    n = (C.c_uint64 * 4)(int(n_tri), int(n_box), 0 if e is None else e.shape[0], 0 if a is None else a.shape[0])
    out = (ShardRange * int(world))()
    flags = (F_BOX_TILED if box_tiled else 0) | (F_TRI_TILED if tri_tiled else 0)
    _check(L.rt_dp_shard_plan(C.byref(n), None if e is None else e.ctypes.data, None if a is None else a.ctypes.data, int(world), flags, out))
    names = ("tri", "box", "euclid", "angular")
    return [{names[op]: (int(r.start[op]), int(r.stop[op])) for op in range(4)} for r in out]


def shard_plan_records(records, world):
    """rt_dp_shard_plan_records on a host record stream ([n, 84] uint32 words).  Returns the world + 1 cut positions."""
    L = load()
    raw = np.ascontiguousarray(records).view(np.uint32).reshape(-1, 84)
    cuts = np.zeros(int(world) + 1, np.uint64)
    _check(L.rt_dp_shard_plan_records(raw.ctypes.data, raw.shape[0], int(world), cuts.ctypes.data))
    return [int(c) for c in cuts]


def knn_tree_build_host(points, leaf_size=128, index_base=0):
    """Native k-d index build on host arrays (numpy float32 [n, 16]).  Returns a dict of numpy arrays:
    points, index, leaf_lo, leaf_hi, split_dim, split_val and n_leaves."""
    L = load()
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 16)
    n = points.shape[0]
    nl = L.rt_dp_knn_tree_leaves(n, leaf_size)
    out = {"n_leaves": nl, "leaf_size": leaf_size,
           "points": np.empty((nl * leaf_size, 16), np.float32), "index": np.empty(nl * leaf_size, np.uint32),
           "leaf_lo": np.empty((nl, 16), np.float32), "leaf_hi": np.empty((nl, 16), np.float32),
           "split_dim": np.zeros(max(nl - 1, 1), np.uint32), "split_val": np.zeros(max(nl - 1, 1), np.float32)}
    _check(L.rt_dp_knn_tree_build_host(points.ctypes.data, n, leaf_size, index_base, out["points"].ctypes.data, out["index"].ctypes.data,
                                       out["leaf_lo"].ctypes.data, out["leaf_hi"].ctypes.data, out["split_dim"].ctypes.data,
                                       out["split_val"].ctypes.data))
    return out


def knn_home_leaf_host(queries, tree):
    L = load()
    queries = np.ascontiguousarray(queries, np.float32).reshape(-1, 16)
    home = np.empty(queries.shape[0], np.uint32)
    _check(L.rt_dp_knn_home_leaf_host(queries.ctypes.data, queries.shape[0], tree["n_leaves"], tree["split_dim"].ctypes.data,
                                      tree["split_val"].ctypes.data, home.ctypes.data))
    return home


def bvh4_build_host(tris, leaf_size=4):
    """Native BVH4 build on host arrays (numpy float32 [n, 9]).  Returns (nodes [m, 32], tris12 [n, 12], order [n])."""
    L = load()
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = tris.shape[0]
    nodes = np.empty((L.rt_dp_bvh4_max_nodes(n), 32), np.float32)
    tris12 = np.empty((n, 12), np.float32)
    order = np.empty(n, np.uint32)
    m = C.c_uint32(0)
    _check(L.rt_dp_bvh4_build_host(tris.ctypes.data, n, leaf_size, nodes.ctypes.data, C.byref(m), tris12.ctypes.data, order.ctypes.data))
    return nodes[:m.value].copy(), tris12, order
