"""ctypes binding of libmi355slam.so (the MI355X hot path of AaltoML/SLAM-module).

This is plumbing for tests and bench.py: every call goes straight through the C ABI declared in
include/mi355slam.h.  There is no CPU fallback here and nothing in this package imports oracle/:
if the HIP library is missing or no gfx950 device is usable, loading / Context() raises.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmi355slam.so")
MAX_LEVELS = 16

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class MsError(RuntimeError):
    pass


class OrbConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("levels", C.c_int32), ("scale_factor", C.c_float),
                ("max_kpts", C.c_int32), ("lk_track_level", C.c_int32), ("fast_threshold", C.c_int32),
                ("max_tracks", C.c_int32), ("max_batch", C.c_int32), ("min_distance", C.c_float)]


class KeypointsView(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("count", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("angle", C.c_void_p), ("octave", C.c_void_p), ("desc", C.c_void_p), ("track_id", C.c_void_p)]


class Bow(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("node_start", C.c_void_p), ("kp_idx", C.c_void_p)]


class MatchFrame(C.Structure):
    _fields_ = [("n", C.c_int32), ("desc", C.c_void_p), ("angle", C.c_void_p), ("octave", C.c_void_p),
                ("bearing", C.c_void_p), ("usable", C.c_void_p), ("bow", Bow)]


_lib = None


def lib():
    """Load the shared library; raise loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsError("libmi355slam.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.ms_last_error.restype = C.c_char_p
        _lib.ms_version.restype = C.c_char_p
        _lib.ms_ctx_stream.restype = C.c_void_p
    return _lib


MS_ERR_TOO_LATE = -6


def prepare_process(concurrent_contexts):
    """ms_prepare_process: before the first HIP call of the process -- two hardware queues per context that will drive the GPU at once (at least 4, at most 32).
    Returns True when the setting is in place (made now, or already in the environment), False when the GPU runtime of this process was up already
    (MS_ERR_TOO_LATE: e.g. under a profiler whose preloaded tool initialises the GPU first) -- the process then runs on the queues it has."""
    rc = lib().ms_prepare_process(int(concurrent_contexts))
    if rc == MS_ERR_TOO_LATE:
        return False
    if rc != 0:
        raise MsError("ms_prepare_process(%d) failed with %d" % (concurrent_contexts, rc))
    return True


def hw_queues():
    """The number of hardware queues the HIP runtime of this process maps its streams onto (GPU_MAX_HW_QUEUES at the time it came up; its default is 4).
    Read from the C environment: ms_prepare_process sets the variable with setenv(), which Python's os.environ snapshot does not see."""
    libc = C.CDLL(None)
    libc.getenv.restype = C.c_char_p
    v = libc.getenv(b"GPU_MAX_HW_QUEUES")
    return int(v) if v else 4


def _vp(x):
    """device pointer / numpy array / None -> c_void_p"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, DevBuf):
        return C.c_void_p(x.ptr)
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    raise TypeError(type(x))


class Context:
    def __init__(self, device=0):
        self._children = []          # weakrefs to objects that must be destroyed before the context
        self._pinned = []            # page-locked host blocks handed out by pinned()
        self._h = C.c_void_p()
        rc = lib().ms_ctx_create(device, C.byref(self._h))
        if rc != 0:
            raise MsError("ms_ctx_create(device=%d) failed with %d: no usable gfx950 device (there is no CPU fallback)" % (device, rc))

    def check(self, rc, what=""):
        if rc != 0:
            raise MsError("%s failed (%d): %s" % (what, rc, lib().ms_last_error(self._h).decode()))

    def sync(self):
        self.check(lib().ms_ctx_sync(self._h), "ms_ctx_sync")

    def stream(self):
        return lib().ms_ctx_stream(self._h)

    def timer_start(self):
        self.check(lib().ms_timer_start(self._h), "ms_timer_start")

    def timer_stop_ms(self):
        ms = C.c_float()
        self.check(lib().ms_timer_stop_ms(self._h, C.byref(ms)), "ms_timer_stop_ms")
        return ms.value

    def event_mark(self, slot):
        self.check(lib().ms_event_mark(self._h, slot), "ms_event_mark")

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self.check(lib().ms_event_elapsed_ms(self._h, a, b, C.byref(ms)), "ms_event_elapsed_ms")
        return ms.value

    def set_hamming_path(self, path):
        """0 = automatic (matrix-core kernel for unmasked searches), 1 = popcount kernel for everything."""
        self.check(lib().ms_hamming_set_path(self._h, int(path)), "ms_hamming_set_path")

    def set_match_path(self, path):
        """0 = node-parallel greedy matchers (default), 1 = one sequential wavefront per keyframe pair."""
        self.check(lib().ms_match_set_path(self._h, int(path)), "ms_match_set_path")

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def pinned(self, shape, dtype=np.uint8):
        """A numpy array in page-locked host memory (ms_host_alloc); freed with the context or by free_pinned()."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self.check(lib().ms_host_alloc(self._h, C.c_size_t(max(n, 1)), C.byref(p)), "ms_host_alloc")
        self._pinned.append(p.value)
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free_pinned(self):
        for p in self._pinned:
            lib().ms_host_free(self._h, C.c_void_p(p))
        self._pinned = []

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DevBuf(self, arr.nbytes)
        b.shape, b.dtype = arr.shape, arr.dtype
        self.check(lib().ms_dev_upload(self._h, C.c_void_p(b.ptr), _vp(arr), C.c_size_t(arr.nbytes)), "ms_dev_upload")
        return b

    def _adopt(self, obj):
        """obj must be destroyed before the context: kept as a weak reference (the list is pruned as it grows -- a problem per frame must not leave a reference per frame)."""
        if len(self._children) > 1024:
            self._children = [r for r in self._children if r() is not None]
        self._children.append(weakref.ref(obj))

    def close(self):
        if self._h:
            for r in self._children:
                o = r()
                if o is not None:
                    o.close()
            self._children = []
            lib().ms_ctx_sync(self._h)
            self.free_pinned()
            lib().ms_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevBuf:
    """A device allocation owned through the C ABI (ms_dev_alloc / ms_dev_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx.check(lib().ms_dev_alloc(ctx._h, C.c_size_t(self.nbytes), C.byref(p)), "ms_dev_alloc")
        ctx._adopt(self)
        self.ptr = p.value or 0
        self.shape, self.dtype = None, None

    def download(self, dtype=None, shape=None):
        dtype = np.dtype(dtype or self.dtype)
        shape = shape if shape is not None else self.shape
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.check(lib().ms_dev_download(self.ctx._h, _vp(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)), "ms_dev_download")
        return out

    def free(self):
        if self.ptr and self.ctx._h:
            lib().ms_dev_free(self.ctx._h, C.c_void_p(self.ptr))
        self.ptr = 0

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- geometry (host-side tables of the product) ----
def scale_factors(levels, f):
    out = np.zeros(levels, np.float32)
    assert lib().ms_scale_factors(levels, C.c_float(f), out.ctypes.data_as(f32p)) == 0
    return out


def level_sigma_sq(levels, f):
    out = np.zeros(levels, np.float32)
    assert lib().ms_level_sigma_sq(levels, C.c_float(f), out.ctypes.data_as(f32p)) == 0
    return out


def level_quotas(levels, f, max_kpts):
    out = np.zeros(levels, np.int32)
    assert lib().ms_level_quotas(levels, C.c_float(f), max_kpts, out.ctypes.data_as(i32p)) == 0
    return out


def level_sizes(levels, f, w, h):
    ws, hs = np.zeros(levels, np.int32), np.zeros(levels, np.int32)
    assert lib().ms_level_sizes(levels, C.c_float(f), w, h, ws.ctypes.data_as(i32p), hs.ctypes.data_as(i32p)) == 0
    return ws, hs


class OrbExtractor:
    """Mirror of slam::OrbExtractor (orb_extractor.hpp:11-30) over the C ABI, batched."""

    def __init__(self, ctx, width, height, levels=8, scale_factor=1.2, max_kpts=2000, lk_track_level=0,
                 fast_threshold=20, max_tracks=0, max_batch=1, min_distance=0.0):
        self.ctx = ctx
        self.cfg = OrbConfig(width, height, levels, scale_factor, max_kpts, lk_track_level, fast_threshold, max_tracks, max_batch, min_distance)
        self._h = C.c_void_p()
        ctx.check(lib().ms_orb_create(ctx._h, C.byref(self.cfg), C.byref(self._h)), "ms_orb_create")
        ctx._adopt(self)
        self.capacity = lib().ms_orb_capacity(self._h)

    def set_valid_mask(self, mask):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.ctx.check(lib().ms_orb_set_valid_mask(self._h, _vp(m)), "ms_orb_set_valid_mask")

    def extract(self, images, n_frames=None, frame_stride=None, row_stride=None, track_xy=None, track_id=None, n_tracks=None):
        """images: numpy uint8 [n,h,w] / [h,w] (host) or a device pointer (int / DevBuf) with explicit strides."""
        on_device = 0
        if isinstance(images, np.ndarray):
            images = np.ascontiguousarray(images, np.uint8)
            if images.ndim == 2:
                images = images[None]
            n_frames = images.shape[0]
            row_stride = images.shape[2]
            frame_stride = images.shape[1] * images.shape[2]
            self._keep = images
        else:
            on_device = 1
            row_stride = row_stride or self.cfg.width
            frame_stride = frame_stride or row_stride * self.cfg.height
        txy = tid = nt = None
        if track_xy is not None:
            T = self.cfg.max_tracks
            txy = np.zeros((n_frames, T, 2), np.float32)
            tid = np.zeros((n_frames, T), np.int32)
            nt = np.zeros(n_frames, np.int32)
            for f in range(n_frames):
                k = len(track_xy[f])
                assert k <= T
                nt[f] = k
                if k:
                    txy[f, :k] = np.asarray(track_xy[f], np.float32).reshape(k, 2)
                    tid[f, :k] = np.asarray(track_id[f], np.int32) if track_id is not None else np.arange(k)
        self.ctx.check(lib().ms_orb_extract(self._h, _vp(images), on_device, n_frames, C.c_size_t(frame_stride),
                                            C.c_size_t(row_stride), _vp(txy), _vp(tid), _vp(nt)), "ms_orb_extract")
        self.n_frames = n_frames

    def device_view(self):
        v = KeypointsView()
        self.ctx.check(lib().ms_orb_device_view(self._h, C.byref(v)), "ms_orb_device_view")
        return v

    def download(self, frame):
        cap = self.capacity
        out = dict(x=np.zeros(cap, np.float32), y=np.zeros(cap, np.float32), angle=np.zeros(cap, np.float32),
                   octave=np.zeros(cap, np.int32), desc=np.zeros((cap, 8), np.uint32), track_id=np.zeros(cap, np.int32))
        n = C.c_int32()
        self.ctx.check(lib().ms_orb_download(self._h, frame, _vp(out["x"]), _vp(out["y"]), _vp(out["angle"]), _vp(out["octave"]),
                                             _vp(out["desc"]), _vp(out["track_id"]), C.byref(n)), "ms_orb_download")
        return {k: v[:n.value].copy() for k, v in out.items()}

    def pack_keypoints(self, frame, bearing=None):
        """KeyPoint::serialize records (76 bytes each: x, y, angle, octave, octave, bearing[3] f64, descriptor[8]) of frame `frame`."""
        self.ctx.sync()
        v = self.device_view()
        cnt = np.zeros(1, np.int32)
        self.ctx.check(lib().ms_dev_download(self.ctx._h, _vp(cnt), C.c_void_p(v.count + 4 * frame), C.c_size_t(4)), "ms_dev_download")
        n = int(cnt[0])
        out = np.zeros((n, 76), np.uint8)
        db = None if bearing is None else self.ctx.upload(np.ascontiguousarray(bearing, np.float64).reshape(n, 3))
        self.ctx.check(lib().ms_keypoints_pack(self.ctx._h, C.byref(v), frame, n, _vp(db), _vp(out)), "ms_keypoints_pack")
        return out

    def set_profiling(self, enable=True):
        self.ctx.check(lib().ms_orb_set_profiling(self._h, int(enable)), "ms_orb_set_profiling")

    def stage_ms(self):
        ms = (C.c_float * 6)()
        self.ctx.check(lib().ms_orb_stage_ms(self._h, ms), "ms_orb_stage_ms")
        return dict(zip(("resize", "blur", "fast", "select", "tracks", "describe"), [float(v) for v in ms]))

    def stage_ms_back(self, calls_back):
        """Stage times of the profiled extract `calls_back` calls ago (0 = the last one; the library keeps 128)."""
        ms = (C.c_float * 6)()
        self.ctx.check(lib().ms_orb_stage_ms_back(self._h, int(calls_back), ms), "ms_orb_stage_ms_back")
        return dict(zip(("resize", "blur", "fast", "select", "tracks", "describe"), [float(v) for v in ms]))

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self.ctx.check(lib().ms_orb_level_size(self._h, level, C.byref(w), C.byref(h)), "ms_orb_level_size")
        return w.value, h.value

    def download_level(self, frame, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self.ctx.check(lib().ms_orb_download_level(self._h, frame, level, int(blurred), _vp(out)), "ms_orb_download_level")
        return out

    def download_detections(self, frame, level):
        q = self.capacity
        x, y, s = np.zeros(q, np.int32), np.zeros(q, np.int32), np.zeros(q, np.int32)
        n = C.c_int32()
        self.ctx.check(lib().ms_orb_download_detections(self._h, frame, level, _vp(x), _vp(y), _vp(s), C.byref(n)), "ms_orb_download_detections")
        return x[:n.value].copy(), y[:n.value].copy(), s[:n.value].copy()

    def close(self):
        if self._h and self.ctx._h:
            lib().ms_orb_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_keypoints(records):
    """Inverse of OrbExtractor.pack_keypoints (host): dict of x, y, angle, octave, bearing, desc."""
    rec = np.ascontiguousarray(records, np.uint8).reshape(-1, 76)
    n = len(rec)
    out = dict(x=np.zeros(n, np.float32), y=np.zeros(n, np.float32), angle=np.zeros(n, np.float32), octave=np.zeros(n, np.int32),
               bearing=np.zeros((n, 3), np.float64), desc=np.zeros((n, 8), np.uint32))
    rc = lib().ms_keypoints_unpack(_vp(rec), n, _vp(out["x"]), _vp(out["y"]), _vp(out["angle"]), _vp(out["octave"]), _vp(out["bearing"]), _vp(out["desc"]))
    if rc != 0:
        raise MsError("ms_keypoints_unpack failed (%d): the two octave fields of a record differ" % rc)
    return out


# ---- matching ----
def hamming_best2(ctx, q, t, n_pairs=1, q_bucket=None, t_bucket=None, t_valid=None):
    """q: [n_pairs*nq, 8] uint32 (numpy, uploaded) -> (best_idx, best_dist, second_dist) numpy arrays."""
    q = np.ascontiguousarray(q, np.uint32).reshape(-1, 8)
    t = np.ascontiguousarray(t, np.uint32).reshape(-1, 8)
    nq, nt = len(q) // n_pairs, len(t) // n_pairs
    dq, dt = ctx.upload(q), ctx.upload(t)
    dqb = ctx.upload(np.asarray(q_bucket, np.int32)) if q_bucket is not None else None
    dtb = ctx.upload(np.asarray(t_bucket, np.int32)) if t_bucket is not None else None
    dtv = ctx.upload(np.asarray(t_valid, np.uint8)) if t_valid is not None else None
    bi, bd, sd = ctx.alloc(4 * nq * n_pairs + 4), ctx.alloc(2 * nq * n_pairs + 4), ctx.alloc(2 * nq * n_pairs + 4)
    ctx.check(lib().ms_hamming_best2(ctx._h, _vp(dq), nq, _vp(dt), nt, n_pairs, _vp(dqb), _vp(dtb), _vp(dtv), _vp(bi), _vp(bd), _vp(sd)),
              "ms_hamming_best2")
    ctx.sync()
    return (bi.download(np.int32, (nq * n_pairs,)), bd.download(np.uint16, (nq * n_pairs,)), sd.download(np.uint16, (nq * n_pairs,)))


def hamming_best2_sets(ctx, q_pool, q_stride, q_count, t_pool, t_stride, t_count, pair_q, pair_t, n_pairs, best_idx, best_dist, second_dist):
    """All arguments are device pointers (int / DevBuf / None); asynchronous on the context stream."""
    ctx.check(lib().ms_hamming_best2_sets(ctx._h, _vp(q_pool), q_stride, _vp(q_count), _vp(t_pool), t_stride, _vp(t_count),
                                          _vp(pair_q), _vp(pair_t), n_pairs, _vp(best_idx), _vp(best_dist), _vp(second_dist)),
              "ms_hamming_best2_sets")


def ratio_test_device(ctx, best_idx, best_dist, second_dist, n, lowe_ratio, max_dist, match):
    ctx.check(lib().ms_ratio_test(ctx._h, _vp(best_idx), _vp(best_dist), _vp(second_dist), n, C.c_float(lowe_ratio), max_dist, _vp(match)),
              "ms_ratio_test")


def ratio_test(ctx, best_idx, best_dist, second_dist, lowe_ratio, max_dist=50):
    n = len(best_idx)
    a, b, c = ctx.upload(np.asarray(best_idx, np.int32)), ctx.upload(np.asarray(best_dist, np.uint16)), ctx.upload(np.asarray(second_dist, np.uint16))
    m = ctx.alloc(4 * n + 4)
    ctx.check(lib().ms_ratio_test(ctx._h, _vp(a), _vp(b), _vp(c), n, C.c_float(lowe_ratio), max_dist, _vp(m)), "ms_ratio_test")
    ctx.sync()
    return m.download(np.int32, (n,))


def feature_search_sort(x, y):
    """FeatureSearch::create: (sorted_x, sorted_y, sorted_idx), stable by y."""
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    sx, sy, si = np.zeros_like(x), np.zeros_like(y), np.zeros(len(x), np.int32)
    rc = lib().ms_feature_search_sort(_vp(x), _vp(y), len(x), _vp(sx), _vp(sy), _vp(si))
    if rc != 0:
        raise MsError("ms_feature_search_sort failed (%d)" % rc)
    return sx, sy, si


def projection_candidates(ctx, kp_x, kp_y, t_desc, q_x, q_y, q_r, q_desc, t_octave=None, t_skip=None, q_min_octave=None, q_max_octave=None):
    """Radius query + best/second scan per query.  Returns (best_idx, best_dist, second_dist, best_oct, second_oct, second_idx, n_candidates)."""
    sx, sy, si = feature_search_sort(kp_x, kp_y)
    t = np.ascontiguousarray(t_desc, np.uint32).reshape(-1, 8); q = np.ascontiguousarray(q_desc, np.uint32).reshape(-1, 8)
    nq, n = len(q), len(sx)
    up = lambda a, dt: ctx.upload(np.ascontiguousarray(a, dt) if len(a) else np.zeros(4, dt))
    dsx, dsy, dsi, dt_, dq = up(sx, np.float32), up(sy, np.float32), up(si, np.int32), ctx.upload(t if n else np.zeros((1, 8), np.uint32)), ctx.upload(q if nq else np.zeros((1, 8), np.uint32))
    doc = None if t_octave is None else up(t_octave, np.int32)
    dsk = None if t_skip is None else up(t_skip, np.uint8)
    dqx, dqy, dqr = up(q_x, np.float32), up(q_y, np.float32), up(q_r, np.float32)
    dlo = None if q_min_octave is None else up(q_min_octave, np.int32)
    dhi = None if q_max_octave is None else up(q_max_octave, np.int32)
    outs = [ctx.alloc(4 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16)]
    ctx.check(lib().ms_projection_candidates(ctx._h, _vp(dsx), _vp(dsy), _vp(dsi), n, _vp(dt_), _vp(doc), _vp(dsk), _vp(dqx), _vp(dqy), _vp(dqr), _vp(dlo), _vp(dhi),
                                             _vp(dq), nq, *[_vp(o) for o in outs]), "ms_projection_candidates")
    ctx.sync()
    dts = (np.int32, np.uint16, np.uint16, np.int32, np.int32, np.int32, np.int32)
    return tuple(o.download(d, (nq,)) for o, d in zip(outs, dts))


def projection_topk(ctx, kp_x, kp_y, t_desc, q_x, q_y, q_r, q_desc, t_octave=None, t_skip=None, q_min_octave=None, q_max_octave=None):
    """Radius query + the four best candidates per query (ms_projection_topk).  Returns (top_idx [nq,4], top_dist [nq,4], top_octave [nq,4], n_scored, n_candidates)."""
    sx, sy, si = feature_search_sort(kp_x, kp_y)
    t = np.ascontiguousarray(t_desc, np.uint32).reshape(-1, 8); q = np.ascontiguousarray(q_desc, np.uint32).reshape(-1, 8)
    nq, n = len(q), len(sx)
    up = lambda a, dt: ctx.upload(np.ascontiguousarray(a, dt) if len(a) else np.zeros(4, dt))
    dsx, dsy, dsi, dt_, dq = up(sx, np.float32), up(sy, np.float32), up(si, np.int32), ctx.upload(t if n else np.zeros((1, 8), np.uint32)), ctx.upload(q if nq else np.zeros((1, 8), np.uint32))
    doc = None if t_octave is None else up(t_octave, np.int32)
    dsk = None if t_skip is None else up(t_skip, np.uint8)
    dqx, dqy, dqr = up(q_x, np.float32), up(q_y, np.float32), up(q_r, np.float32)
    dlo = None if q_min_octave is None else up(q_min_octave, np.int32)
    dhi = None if q_max_octave is None else up(q_max_octave, np.int32)
    ti, td, to, ns, nc = ctx.alloc(16 * nq + 16), ctx.alloc(8 * nq + 16), ctx.alloc(16 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16)
    ctx.check(lib().ms_projection_topk(ctx._h, _vp(dsx), _vp(dsy), _vp(dsi), n, _vp(dt_), _vp(doc), _vp(dsk), _vp(dqx), _vp(dqy), _vp(dqr), _vp(dlo), _vp(dhi),
                                       _vp(dq), nq, _vp(ti), _vp(td), _vp(to), _vp(ns), _vp(nc)), "ms_projection_topk")
    ctx.sync()
    return (ti.download(np.int32, (nq, 4)), td.download(np.uint16, (nq, 4)), to.download(np.int32, (nq, 4)), ns.download(np.int32, (nq,)), nc.download(np.int32, (nq,)))


def hamming_candidates_topk(ctx, q_desc, t_desc, cand_lists, t_skip=None, t_octave=None):
    """The four best of each query's own candidate list (ms_hamming_candidates_topk): (top_idx [nq,4], top_dist [nq,4], top_octave [nq,4], n_scored)."""
    q = np.ascontiguousarray(q_desc, np.uint32).reshape(-1, 8); t = np.ascontiguousarray(t_desc, np.uint32).reshape(-1, 8)
    nq = len(q)
    start = np.zeros(nq + 1, np.int32)
    start[1:] = np.cumsum([len(c) for c in cand_lists])
    idx = np.concatenate([np.asarray(c, np.int32) for c in cand_lists] + [np.zeros(0, np.int32)]).astype(np.int32)
    dq, dt, ds, di = ctx.upload(q), ctx.upload(t if len(t) else np.zeros((1, 8), np.uint32)), ctx.upload(start), ctx.upload(idx if len(idx) else np.zeros(1, np.int32))
    dsk = ctx.upload(np.asarray(t_skip, np.uint8)) if t_skip is not None else None
    doc = ctx.upload(np.asarray(t_octave, np.int32)) if t_octave is not None else None
    ti, td, to, ns = ctx.alloc(16 * nq + 16), ctx.alloc(8 * nq + 16), ctx.alloc(16 * nq + 16), ctx.alloc(4 * nq + 16)
    ctx.check(lib().ms_hamming_candidates_topk(ctx._h, _vp(dq), nq, _vp(dt), _vp(ds), _vp(di), _vp(dsk), _vp(doc), _vp(ti), _vp(td), _vp(to), _vp(ns)), "ms_hamming_candidates_topk")
    ctx.sync()
    return ti.download(np.int32, (nq, 4)), td.download(np.uint16, (nq, 4)), to.download(np.int32, (nq, 4)), ns.download(np.int32, (nq,))


def descriptor_medoid(ctx, desc_pool, obs_lists):
    """MapPoint::updateDescriptor for many map points: obs_lists[p] = indices into desc_pool.  Returns (best_local, best_pool)."""
    pool = np.ascontiguousarray(desc_pool, np.uint32).reshape(-1, 8)
    n = len(obs_lists)
    start = np.zeros(n + 1, np.int32)
    start[1:] = np.cumsum([len(o) for o in obs_lists])
    idx = np.concatenate([np.asarray(o, np.int32) for o in obs_lists] + [np.zeros(0, np.int32)]).astype(np.int32)
    dp, ds, di = ctx.upload(pool if len(pool) else np.zeros((1, 8), np.uint32)), ctx.upload(start), ctx.upload(idx if len(idx) else np.zeros(1, np.int32))
    bl, bp = ctx.alloc(4 * n + 16), ctx.alloc(4 * n + 16)
    ctx.check(lib().ms_descriptor_medoid(ctx._h, _vp(dp), _vp(ds), _vp(di), n, max([len(o) for o in obs_lists] + [0]), _vp(bl), _vp(bp)),
              "ms_descriptor_medoid")
    ctx.sync()
    return bl.download(np.int32, (n,)), bp.download(np.int32, (n,))


class BowVocabulary:
    """Device copy of a DBoW2 vocabulary tree (ms_bow_vocab): parent ids, node descriptors, weights, word ids, depth L."""

    def __init__(self, ctx, parent, node_desc, node_weight, node_word, depth_levels):
        self.ctx = ctx
        par = np.ascontiguousarray(parent, np.int32); nd = np.ascontiguousarray(node_desc, np.uint32).reshape(-1, 8)
        wt = np.ascontiguousarray(node_weight, np.float64); wd = np.ascontiguousarray(node_word, np.int32)
        if not (len(par) == len(nd) == len(wt) == len(wd)): raise ValueError("vocabulary arrays differ in length")
        h = C.c_void_p()
        ctx.check(lib().ms_bow_vocab_create(ctx._h, len(par), par.ctypes.data_as(C.c_void_p), nd.ctypes.data_as(C.c_void_p),
                                            wt.ctypes.data_as(C.c_void_p), wd.ctypes.data_as(C.c_void_p), int(depth_levels), C.byref(h)), "ms_bow_vocab_create")
        self._h = h
        ctx._adopt(self)

    def transform(self, desc, levels_up=4):
        """desc: host array [n, 8] u32 or a device buffer (then pass n via a tuple (buf, n)).  Returns word, weight, node (host arrays)."""
        if isinstance(desc, tuple): buf, n = desc
        else:
            d = np.ascontiguousarray(desc, np.uint32).reshape(-1, 8); n = len(d)
            buf = self.ctx.upload(d if n else np.zeros((1, 8), np.uint32))
        w, wt, nd = self.ctx.alloc(4 * n + 16), self.ctx.alloc(8 * n + 16), self.ctx.alloc(4 * n + 16)
        self.ctx.check(lib().ms_bow_transform(self.ctx._h, self._h, _vp(buf), n, int(levels_up), _vp(w), _vp(wt), _vp(nd)), "ms_bow_transform")
        self.ctx.sync()
        return w.download(np.int32, (n,)), wt.download(np.float64, (n,)), nd.download(np.int32, (n,))

    def close(self):
        if getattr(self, "_h", None):
            lib().ms_bow_vocab_destroy(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass


def hamming_candidates(ctx, q_desc, t_desc, cand_lists, t_skip=None, t_octave=None):
    """cand_lists: list (per query) of keypoint index arrays.  Returns (best_idx, best_dist, second_dist, best_oct, second_oct)."""
    q = np.ascontiguousarray(q_desc, np.uint32).reshape(-1, 8); t = np.ascontiguousarray(t_desc, np.uint32).reshape(-1, 8)
    nq = len(q)
    start = np.zeros(nq + 1, np.int32)
    start[1:] = np.cumsum([len(c) for c in cand_lists])
    idx = np.concatenate([np.asarray(c, np.int32) for c in cand_lists] + [np.zeros(0, np.int32)]).astype(np.int32)
    dq, dt, ds, di = ctx.upload(q), ctx.upload(t if len(t) else np.zeros((1, 8), np.uint32)), ctx.upload(start), ctx.upload(idx if len(idx) else np.zeros(1, np.int32))
    dsk = ctx.upload(np.asarray(t_skip, np.uint8)) if t_skip is not None else None
    doc = ctx.upload(np.asarray(t_octave, np.int32)) if t_octave is not None else None
    outs = [ctx.alloc(4 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16)]
    ctx.check(lib().ms_hamming_candidates(ctx._h, _vp(dq), nq, _vp(dt), _vp(ds), _vp(di), _vp(dsk), _vp(doc), *[_vp(o) for o in outs], None), "ms_hamming_candidates")
    ctx.sync()
    return (outs[0].download(np.int32, (nq,)), outs[1].download(np.uint16, (nq,)), outs[2].download(np.uint16, (nq,)),
            outs[3].download(np.int32, (nq,)), outs[4].download(np.int32, (nq,)))


def angle_check(delta, ids):
    delta = np.ascontiguousarray(delta, np.float32); ids = np.ascontiguousarray(ids, np.int32)
    inv = np.zeros(max(len(ids), 1), np.int32)
    m = lib().ms_angle_check(delta.ctypes.data_as(f32p), ids.ctypes.data_as(i32p), len(ids), inv.ctypes.data_as(i32p))
    assert m >= 0
    return inv[:m].copy()


class FrameOnDevice:
    """Uploads one keyframe's matcher inputs and builds the ms_match_frame struct."""

    def __init__(self, ctx, desc, angle, usable, bucket, octave=None, bearing=None, csr=None):
        """bucket: vocabulary node of every keypoint (the CSR is built like DBoW2 fills a FeatureVector: nodes ascending, keypoints of a
        node in index order); csr = (node_id, node_start, kp_idx) overrides it with explicit node lists (tests)."""
        if csr is None:
            bucket = np.asarray(bucket, np.int32)
            order = np.argsort(bucket, kind="stable").astype(np.int32)
            ids, counts = np.unique(bucket, return_counts=True)
            start = np.zeros(len(ids) + 1, np.int32)
            start[1:] = np.cumsum(counts)
        else:
            ids, start, order = [np.ascontiguousarray(a, np.int32) for a in csr]
        self.n = len(np.asarray(usable))
        self.bufs = dict(desc=ctx.upload(np.asarray(desc, np.uint32).reshape(-1, 8)), angle=ctx.upload(np.asarray(angle, np.float32)),
                         usable=ctx.upload(np.asarray(usable, np.uint8)), node_id=ctx.upload(ids.astype(np.int32)),
                         node_start=ctx.upload(start), kp_idx=ctx.upload(order))
        if octave is not None:
            self.bufs["octave"] = ctx.upload(np.asarray(octave, np.int32))
        if bearing is not None:
            self.bufs["bearing"] = ctx.upload(np.asarray(bearing, np.float64).reshape(-1, 3))
        b = self.bufs
        self.struct = MatchFrame(self.n, b["desc"].ptr, b["angle"].ptr, b["octave"].ptr if "octave" in b else 0,
                                 b["bearing"].ptr if "bearing" in b else 0, b["usable"].ptr,
                                 Bow(len(ids), b["node_id"].ptr, b["node_start"].ptr, b["kp_idx"].ptr))


def _run_greedy(ctx, frames1, frames2, call):
    n = len(frames1)
    A1 = (MatchFrame * n)(*[f.struct for f in frames1])
    A2 = (MatchFrame * n)(*[f.struct for f in frames2])
    outs = [ctx.alloc(4 * max(f.n, 1)) for f in frames1]
    ptrs = (C.c_void_p * n)(*[o.ptr for o in outs])
    nm = ctx.alloc(4 * n)
    call(A1, A2, n, ptrs, nm)
    ctx.sync()
    counts = nm.download(np.int32, (n,))
    return counts, [o.download(np.int32, (f.n,)) for o, f in zip(outs, frames1)]


def match_loop_closure(ctx, frames1, frames2, lowe_ratio, check_orientation=True):
    def call(A1, A2, n, ptrs, nm):
        ctx.check(lib().ms_match_loop_closure(ctx._h, A1, A2, n, C.c_float(lowe_ratio), int(check_orientation), ptrs, _vp(nm)),
                  "ms_match_loop_closure")
    return _run_greedy(ctx, frames1, frames2, call)


def match_triangulation(ctx, frames1, frames2, E12, scale_factors_, thr_deg, check_orientation=True):
    dE = ctx.upload(np.asarray(E12, np.float64).reshape(-1, 9))
    dsf = ctx.upload(np.asarray(scale_factors_, np.float32))

    def call(A1, A2, n, ptrs, nm):
        ctx.check(lib().ms_match_triangulation(ctx._h, A1, A2, n, _vp(dE), _vp(dsf), C.c_float(thr_deg), int(check_orientation), ptrs, _vp(nm)),
                  "ms_match_triangulation")
    return _run_greedy(ctx, frames1, frames2, call)


# ---- bundle adjustment ----
class BaProblemC(C.Structure):
    _fields_ = [("n_pose", C.c_int32), ("n_point", C.c_int32), ("n_obs", C.c_int32), ("n_pose_edge", C.c_int32),
                ("pose", C.c_void_p), ("pose_fixed", C.c_void_p), ("point", C.c_void_p), ("point_fixed", C.c_void_p),
                ("obs_pose", C.c_void_p), ("obs_point", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_info", C.c_void_p),
                ("huber_delta", C.c_double), ("edge_i", C.c_void_p), ("edge_j", C.c_void_p), ("edge_meas", C.c_void_p),
                ("edge_info", C.c_void_p), ("max_iters", C.c_int32)]


class BaResultC(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("stopped_early", C.c_int32), ("final_lambda", C.c_double),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("phase_cycles", C.c_double * 8)]


def _ba_struct(prob, max_iters):
    """prob: dict of numpy arrays (see tests/ba_synth.py).  Returns (struct, keepalive)."""
    k = dict(pose=np.ascontiguousarray(prob["pose"], np.float64), pf=np.ascontiguousarray(prob["pose_fixed"], np.uint8),
             point=np.ascontiguousarray(prob["point"], np.float64),
             lf=None if prob.get("point_fixed") is None else np.ascontiguousarray(prob["point_fixed"], np.uint8),
             op=np.ascontiguousarray(prob["obs_pose"], np.int32), ol=np.ascontiguousarray(prob["obs_point"], np.int32),
             uv=np.ascontiguousarray(prob["obs_uv"], np.float64), info=np.ascontiguousarray(prob["obs_info"], np.float64),
             ei=np.ascontiguousarray(prob["edge_i"], np.int32), ej=np.ascontiguousarray(prob["edge_j"], np.int32),
             em=np.ascontiguousarray(prob["edge_meas"], np.float64), ew=np.ascontiguousarray(prob["edge_info"], np.float64))
    a = lambda x: None if x is None else x.ctypes.data
    s = BaProblemC(len(k["pose"]), len(k["point"]), len(k["op"]), len(k["ei"]), a(k["pose"]), a(k["pf"]), a(k["point"]), a(k["lf"]),
                   a(k["op"]), a(k["ol"]), a(k["uv"]), a(k["info"]), float(prob["huber_delta"]), a(k["ei"]), a(k["ej"]), a(k["em"]), a(k["ew"]), max_iters)
    return s, k


class BundleAdjuster:
    """A batch of independent BA problems resident on the device (ms_ba_create / ms_ba_solve / ms_ba_download)."""

    def __init__(self, ctx, problems, max_iters=10):
        self.ctx = ctx
        structs, self._keep = zip(*[_ba_struct(p, max_iters) for p in problems])
        self.n = len(structs)
        self.dims = [(s.n_pose, s.n_point, s.n_obs) for s in structs]
        arr = (BaProblemC * self.n)(*structs)
        self._h = C.c_void_p()
        ctx.check(lib().ms_ba_create(ctx._h, arr, self.n, C.byref(self._h)), "ms_ba_create")
        ctx._adopt(self)

    def set_team(self, workgroups_per_problem):
        self.ctx.check(lib().ms_ba_set_team(self._h, int(workgroups_per_problem)), "ms_ba_set_team")

    def set_factor_team(self, workgroups):
        self.ctx.check(lib().ms_ba_set_factor_team(self._h, int(workgroups)), "ms_ba_set_factor_team")

    def solve(self):
        self.ctx.check(lib().ms_ba_solve(self._h), "ms_ba_solve")

    def copy_state_from(self, src, extra_pose_src=None):
        """The state src's last solve left becomes this handle's initial state (ms_ba_copy_state)."""
        ex = None if extra_pose_src is None else np.ascontiguousarray(extra_pose_src, np.int32)
        self.ctx.check(lib().ms_ba_copy_state(self._h, src._h, _vp(ex)), "ms_ba_copy_state")

    def team_fallbacks(self):
        return lib().ms_ba_team_fallbacks(self._h)

    def debug_force_reject(self, first_trials):
        self.ctx.check(lib().ms_ba_debug_force_reject(self._h, int(first_trials)), "ms_ba_debug_force_reject")

    def debug_fail_team_barriers(self, on=True):
        self.ctx.check(lib().ms_ba_debug_fail_team_barriers(self._h, int(on)), "ms_ba_debug_fail_team_barriers")

    def download(self, i):
        npz, nl, no = self.dims[i]
        pose, point, chi2 = np.zeros((npz, 7)), np.zeros((nl, 3)), np.zeros(no)
        r = BaResultC()
        self.ctx.check(lib().ms_ba_download(self._h, i, _vp(pose), _vp(point), _vp(chi2), C.byref(r)), "ms_ba_download")
        return dict(pose=pose, point=point, chi2=chi2,
                    stats=dict(iters=r.iterations, trials=r.trials, stop=r.stopped_early, lam=r.final_lambda, chi2_init=r.chi2_initial, chi2_final=r.chi2_final,
                               phase_cycles=dict(zip(("eval", "linearise", "schur", "cholesky", "points_update", "total", "schur_init", "schur_wait"), list(r.phase_cycles)))))

    def close(self):
        if self._h and self.ctx._h:
            lib().ms_ba_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
