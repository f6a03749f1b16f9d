"""ctypes binding of libfrp.so (C ABI: include/frp.h).  No CPU fallback: if the library
or a GPU is missing every constructor raises -- the product path never routes around HIP."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Two handles that keep two batches
# in flight on one GPU (lanes.py) need their compute streams on DIFFERENT queues; next to torch's and RCCL's streams four
# queues were not enough (measured: no gain from the second lane in 3 of 3 processes with 4 queues, the full gain in 5 of
# 5 with 8).  The runtime reads the variable when it initialises, so this only helps when this module is imported before
# the first HIP call of the process - export it in the service's environment otherwise.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# FRP_LIB: another build of the same library (same-box A/B of kernel variants: tools/ab_lib.sh)
LIB_PATH = os.environ.get("FRP_LIB") or os.path.join(_HERE, "libfrp.so")

EMB_DIM = 512
CHIP = 112
MAX_FACES_CAP = 128
MAX_TOPK = 64
FLAG_FORCED_K, FLAG_RGB, FLAG_NO_MATCH = 1, 2, 4
F32, F16, F64 = 0, 1, 2

ABI_SYMBOLS = [
    "frp_create", "frp_destroy", "frp_last_error", "frp_version", "frp_load_weights",
    "frp_gallery_set", "frp_gallery_set_device", "frp_gallery_reserve", "frp_gallery_commit", "frp_gallery_cancel", "frp_gallery_device_ptr", "frp_gallery_update_row", "frp_gallery_remove_row",
    "frp_jpeg_info_get", "frp_jpeg_coefficients", "frp_upload_jpeg_async", "frp_debug_jpeg_device_batches", "frp_debug_graph_replays",
    "frp_dist_unique_id", "frp_dist_init", "frp_dist_destroy", "frp_gallery_allgather",
    "frp_gallery_size", "frp_gallery_get", "frp_gallery_exact", "frp_gallery_distances", "frp_gallery_get_exact",
    "frp_process_frames", "frp_upload_frames", "frp_process_resident", "frp_fetch_results", "frp_synchronize",
    "frp_host_alloc", "frp_host_free", "frp_upload_frames_async", "frp_swap_frames",
    "frp_detect", "frp_detect_resident", "frp_get_det_source", "frp_finish_faces", "frp_get_head_map", "frp_debug_det_prefix", "frp_debug_det_hashes", "frp_decode_heads", "frp_align", "frp_embed_aligned", "frp_embed_faces",
    "frp_match", "frp_match_scores", "frp_conv2d_nhwc", "frp_conv2d_f8", "frp_get_counters", "frp_reset_counters", "frp_set_profile",
]


class FrpConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("max_batch", C.c_int32), ("max_faces", C.c_int32),
                ("max_h", C.c_int32), ("max_w", C.c_int32), ("profile", C.c_int32), ("reserved", C.c_int32 * 10)]


class FrpCounters(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("calls", C.c_int32), ("frames", C.c_int64), ("faces", C.c_int64),
                ("ms_h2d", C.c_double), ("ms_preprocess", C.c_double), ("ms_det_conv", C.c_double),
                ("ms_decode", C.c_double), ("ms_align", C.c_double), ("ms_emb_conv", C.c_double),
                ("ms_l2norm", C.c_double), ("ms_match", C.c_double), ("ms_d2h", C.c_double), ("ms_total", C.c_double),
                ("det_conv_flops", C.c_double), ("emb_conv_flops", C.c_double),
                ("det_conv_launches", C.c_int64), ("emb_conv_launches", C.c_int64),
                ("match_bytes", C.c_double), ("match_launches", C.c_int64), ("gallery_rows", C.c_int64),
                ("f8_conv_flops", C.c_double), ("f8_conv_launches", C.c_int64), ("reserved", C.c_double * 6)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n not in ("reserved", "struct_size")}


class JpegInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32), ("h_samp", C.c_int32 * 3), ("v_samp", C.c_int32 * 3),
                ("mcus_x", C.c_int32), ("mcus_y", C.c_int32), ("restart_interval", C.c_int32), ("progressive", C.c_int32)]

    def as_dict(self) -> dict:
        return {"width": self.width, "height": self.height, "components": self.components, "h_samp": list(self.h_samp),
                "v_samp": list(self.v_samp), "mcus_x": self.mcus_x, "mcus_y": self.mcus_y, "restart_interval": self.restart_interval}


def _byte_ptr(data):
    """(pointer, keep-alive) to the bytes of `data` without copying them where the object allows it (bytes and its subclasses:
    the object's own buffer; a bytearray: its buffer; anything else is copied once).  A 1080p JPEG is ~0.7 MB and a batch holds
    32 of them: per-call copies were tens of megabytes of memcpy on the thread that feeds the lane."""
    if isinstance(data, bytes):
        p = C.c_char_p(data)
        return C.cast(p, C.c_void_p), (p, data)
    if isinstance(data, bytearray):
        buf = (C.c_char * len(data)).from_buffer(data)
        return C.cast(buf, C.c_void_p), buf
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    return C.cast(buf, C.c_void_p), buf


def jpeg_info(data: bytes):
    """header of a baseline JPEG as the library's decoder sees it, or None when it does not cover the file (progressive,
    arithmetic-coded, 12-bit, CMYK, multi-scan): include/frp.h frp_jpeg_info_get.  Needs no GPU."""
    info = JpegInfo()
    ptr, keep = _byte_ptr(data)
    rc = load_library().frp_jpeg_info_get(ptr, len(data), C.byref(info))
    del keep
    return info.as_dict() if rc == 0 else None


def jpeg_coefficients(data: bytes):
    """host entropy decode alone (parity tests): -> (info dict, int16 coefficients, uint16 [3, 64] tables)"""
    info = jpeg_info(data)
    if info is None:
        raise FrpError(-1, "not a baseline JPEG the decoder covers")
    n = sum(info["mcus_x"] * info["h_samp"][c] * info["mcus_y"] * info["v_samp"][c] * 64 for c in range(info["components"]))
    coef = np.zeros(n, np.int16)
    q = np.zeros((3, 64), np.uint16)
    ji = JpegInfo()
    ptr, keep = _byte_ptr(data)
    rc = load_library().frp_jpeg_coefficients(ptr, len(data), _ptr(coef), n, _ptr(q), C.byref(ji))
    del keep
    if rc != 0:
        raise FrpError(rc, "corrupt JPEG scan data")
    return info, coef, q


class FrpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"frp error {code}: {msg}")
        self.code = code


_lib = None


def kernel_source_hash() -> str:
    """sha256 over the native sources (csrc/*.hip|h|cpp, include/*.h): ties a committed profile summary
    (profiles/*/pmc_traffic.json) to the kernels it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_library() -> C.CDLL:
    """dlopen libfrp.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `make -C {os.path.join(_HERE, 'csrc')}`")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
    lib.frp_create.argtypes = [C.c_int, C.POINTER(FrpConfig), C.POINTER(vp)]
    lib.frp_destroy.argtypes = [vp]
    lib.frp_destroy.restype = None
    lib.frp_last_error.argtypes = [vp]
    lib.frp_last_error.restype = C.c_char_p
    lib.frp_version.restype = C.c_char_p
    lib.frp_load_weights.argtypes = [vp, vp, C.c_size_t]
    lib.frp_gallery_set.argtypes = [vp, vp, i64, i32, i32]
    lib.frp_gallery_set_device.argtypes = [vp, vp, i64, i32]
    lib.frp_gallery_reserve.argtypes = [vp, i64, C.POINTER(C.c_void_p)]
    lib.frp_gallery_commit.argtypes = [vp, i64]
    lib.frp_gallery_cancel.argtypes = [vp]
    lib.frp_gallery_device_ptr.argtypes = [vp]
    lib.frp_gallery_device_ptr.restype = vp
    lib.frp_gallery_update_row.argtypes = [vp, i64, vp, i32, i32]
    lib.frp_gallery_remove_row.argtypes = [vp, i64]
    lib.frp_gallery_size.argtypes = [vp]
    lib.frp_gallery_size.restype = i64
    lib.frp_gallery_get.argtypes = [vp, vp, i64, i64]
    lib.frp_jpeg_info_get.argtypes = [vp, C.c_size_t, vp]
    lib.frp_jpeg_coefficients.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp]
    lib.frp_upload_jpeg_async.argtypes = [vp, vp, vp, i32]
    lib.frp_gallery_exact.argtypes = [vp, i32]
    lib.frp_gallery_distances.argtypes = [vp, vp, i32, vp, i64]
    lib.frp_gallery_get_exact.argtypes = [vp, vp, i64, i64]
    lib.frp_process_frames.argtypes = [vp, vp, i32, i32, i32, i64, i32, f32, f32, u32, vp, vp, vp, vp, vp, vp, vp]
    lib.frp_upload_frames.argtypes = [vp, vp, i32, i32, i32, i64]
    lib.frp_process_resident.argtypes = [vp, i32, f32, f32, u32]
    lib.frp_fetch_results.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.frp_synchronize.argtypes = [vp]
    lib.frp_host_alloc.argtypes = [vp, C.c_size_t]
    lib.frp_host_alloc.restype = vp
    lib.frp_host_free.argtypes = [vp, vp]
    lib.frp_host_free.restype = None
    lib.frp_upload_frames_async.argtypes = [vp, vp, i32, i32, i32, i64]
    lib.frp_swap_frames.argtypes = [vp]
    lib.frp_detect.argtypes = [vp, vp, i32, i32, i32, i64, i32, f32, f32, u32, vp, vp, vp, vp, vp]
    lib.frp_detect_resident.argtypes = [vp, i32, i32, i32, i32, f32, f32, u32, vp, vp, vp, vp, vp]
    lib.frp_get_det_source.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    lib.frp_finish_faces.argtypes = [vp, i32, vp, vp, vp, vp, i32, u32, vp, vp, vp]
    lib.frp_get_head_map.argtypes = [vp, i32, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    lib.frp_debug_jpeg_device_batches.argtypes = [vp]
    lib.frp_debug_jpeg_device_batches.restype = C.c_int64
    lib.frp_debug_graph_replays.argtypes = [vp]
    lib.frp_debug_graph_replays.restype = C.c_int64
    lib.frp_dist_unique_id.argtypes = [vp]
    lib.frp_dist_init.argtypes = [vp, vp, i32, i32]
    lib.frp_dist_destroy.argtypes = [vp]
    lib.frp_gallery_allgather.argtypes = [vp, vp, i64, i32, i64]
    lib.frp_debug_det_prefix.argtypes = [vp, i32, vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.frp_debug_det_hashes.argtypes = [vp, i32, vp]
    lib.frp_decode_heads.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp, vp, vp, vp, vp]
    lib.frp_align.argtypes = [vp, vp, i32, i32, i64, vp, i32, u32, vp]
    lib.frp_embed_aligned.argtypes = [vp, vp, i32, vp]
    lib.frp_embed_faces.argtypes = [vp, vp, i32, i32, i64, vp, i32, u32, vp]
    lib.frp_match.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.frp_match_scores.argtypes = [vp, vp, i32, vp, i64]
    lib.frp_conv2d_nhwc.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.frp_conv2d_f8.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp]
    if hasattr(lib, "frp_kstep_lab"):            # the FRP_LAB build (libfrp_lab.so, include/frp_lab.h): tuning hooks
        lib.frp_conv_bench.argtypes = [vp] + [i32] * 11 + [C.POINTER(C.c_float), vp]
        lib.frp_mfma_peak.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
        lib.frp_kstep_lab.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    lib.frp_get_counters.argtypes = [vp, C.POINTER(FrpCounters)]
    lib.frp_reset_counters.argtypes = [vp]
    lib.frp_set_profile.argtypes = [vp, i32]
    _lib = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One handle = one GPU + one stream (include/frp.h).  Thread-safe (library mutex);
    ctypes releases the GIL for the duration of every call."""

    def __init__(self, device: int = 0, max_batch: int = 32, max_faces: int = 10, max_h: int = 1080,
                 max_w: int = 1920, profile: bool = False):
        self._lib = load_library()
        cfg = FrpConfig()
        cfg.struct_size = C.sizeof(FrpConfig)
        cfg.max_batch, cfg.max_faces, cfg.max_h, cfg.max_w, cfg.profile = max_batch, max_faces, max_h, max_w, int(profile)
        h = C.c_void_p()
        rc = self._lib.frp_create(device, C.byref(cfg), C.byref(h))
        if rc != 0 or not h.value:
            raise FrpError(rc, "frp_create failed (no usable HIP device?)")
        self._h = h
        self.device = device
        self.max_faces = max_faces
        # Multi-call sequences (upload -> process -> fetch, the pyramid) leave state on the handle between calls:
        # threads sharing one Engine take this lock around a whole sequence (`with eng.sequence(): ...`).  The
        # C side additionally checks every caller-sized buffer against the handle's state under its own mutex.
        self._seq = threading.RLock()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.frp_destroy(self._h)
            self._h = C.c_void_p()

    def sequence(self):
        """lock held around a multi-call sequence on a shared Engine"""
        return self._seq

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise FrpError(rc, (self._lib.frp_last_error(self._h) or b"").decode("utf-8", "replace"))

    # -- weights / gallery
    def load_weights(self, blob: bytes):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._chk(self._lib.frp_load_weights(self._h, C.cast(buf, C.c_void_p), len(blob)))

    def gallery_set(self, emb: np.ndarray):
        emb = np.ascontiguousarray(emb)
        if emb.size == 0:
            self._chk(self._lib.frp_gallery_set(self._h, None, 0, EMB_DIM, F32))
            return
        dt = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.float64): F64}.get(emb.dtype)
        if dt is None:
            emb, dt = emb.astype(np.float32), F32
        if emb.ndim == 2 and emb.shape[1] < EMB_DIM:          # narrower rows (128-d): zero-padded
            emb = np.ascontiguousarray(np.concatenate([emb, np.zeros((emb.shape[0], EMB_DIM - emb.shape[1]), emb.dtype)], axis=1))
        self._chk(self._lib.frp_gallery_set(self._h, _ptr(emb), emb.shape[0], emb.shape[1], dt))

    def gallery_set_device(self, dev_ptr: int, n: int):
        self._chk(self._lib.frp_gallery_set_device(self._h, C.c_void_p(dev_ptr), n, EMB_DIM))

    # -- multi-GPU: the gallery all-gather on the library's own RCCL communicator (include/frp.h: frp_dist_*)
    @staticmethod
    def dist_unique_id() -> bytes:
        """rank 0: a fresh 128-byte communicator id (carry it to the other ranks by any control channel)"""
        buf = C.create_string_buffer(128)
        rc = load_library().frp_dist_unique_id(buf)
        if rc != 0:
            raise FrpError(rc, "frp_dist_unique_id failed (librccl not loadable?)")
        return buf.raw

    def dist_init(self, unique_id: bytes, rank: int, world: int):
        """collective: this handle's communicator"""
        if len(unique_id) != 128:
            raise ValueError("the communicator id has 128 bytes")
        self._chk(self._lib.frp_dist_init(self._h, C.c_char_p(unique_id), rank, world))
        self._dist = (rank, world)

    def dist_destroy(self):
        self._chk(self._lib.frp_dist_destroy(self._h))
        self._dist = None

    def gallery_allgather(self, shard: np.ndarray, n_total: int):
        """collective: this rank's rows (host [rows, 512], any float dtype) -> the full unit fp16 gallery on every rank"""
        shard = np.ascontiguousarray(shard)
        if shard.dtype not in (np.float32, np.float16, np.float64):
            shard = shard.astype(np.float32)
        code = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.float64): 2}[shard.dtype]
        shard = shard.reshape(-1, 512)
        self._chk(self._lib.frp_gallery_allgather(self._h, _ptr(shard) if shard.shape[0] else None, shard.shape[0], code, n_total))

    def gallery_reserve(self, capacity_rows: int) -> int:
        """device address of a fresh, not yet visible snapshot of capacity_rows x 512 fp16 (fill it, then gallery_commit)"""
        ptr = C.c_void_p()
        self._chk(self._lib.frp_gallery_reserve(self._h, capacity_rows, C.byref(ptr)))
        return int(ptr.value)

    def gallery_commit(self, n_rows: int):
        self._chk(self._lib.frp_gallery_commit(self._h, n_rows))

    def gallery_cancel(self):
        """discard a pending reservation (other gallery updates are refused while one is pending)"""
        self._chk(self._lib.frp_gallery_cancel(self._h))

    def gallery_device_ptr(self) -> int:
        """device address of the current snapshot (0 when empty); valid until the next gallery update"""
        return int(self._lib.frp_gallery_device_ptr(self._h) or 0)

    @staticmethod
    def _row512(emb: np.ndarray) -> Tuple[np.ndarray, int]:
        """one embedding as the library takes it: 512 float32 or - a float64 input stays float64, so the exact compat rows
        (gallery_exact) hold it bit for bit - float64 values; narrower rows (the reference's 128-d dlib encodings) are
        zero-padded, which changes neither a cosine nor a Euclidean distance"""
        e = np.asarray(emb)
        dt, code = (np.float64, F64) if e.dtype == np.float64 else (np.float32, F32)
        e = np.ascontiguousarray(e, dtype=dt).reshape(-1)
        if e.shape[0] < EMB_DIM:
            e = np.concatenate([e, np.zeros(EMB_DIM - e.shape[0], dt)])
        return e, code

    def gallery_update_row(self, row: int, emb: np.ndarray):
        e, code = self._row512(emb)
        self._chk(self._lib.frp_gallery_update_row(self._h, row, _ptr(e), e.shape[0], code))

    def upload_jpeg_async(self, jpegs: Sequence[bytes]):
        """B baseline JPEG stills of one geometry -> the staging frame buffer, decoded on the way: the bit streams on host
        threads, dequantisation / inverse DCT / chroma upsampling / YCbCr -> BGR on the GPU's copy stream (frp.h:
        frp_upload_jpeg_async).  Follow with swap_frames() as after upload_frames_async()."""
        B = len(jpegs)
        held = [_byte_ptr(j) for j in jpegs]                      # no copies: the call reads the callers' buffers (and returns after it has)
        ptrs = (C.c_void_p * B)(*[h[0] for h in held])
        sizes = (C.c_size_t * B)(*[len(j) for j in jpegs])
        self._chk(self._lib.frp_upload_jpeg_async(self._h, ptrs, sizes, B))
        del held
        info = jpeg_info(jpegs[0])
        self._staged = (B, info["height"], info["width"])

    def jpeg_device_batches(self) -> int:
        """diagnostic: upload_jpeg_async batches whose entropy decode ran on the device (restart-interval streams)"""
        return int(self._lib.frp_debug_jpeg_device_batches(self._h))

    def graph_replays(self) -> int:
        """diagnostic: detector / embedder passes replayed from a captured hipGraph so far (frp.h: frp_debug_graph_replays)"""
        return int(self._lib.frp_debug_graph_replays(self._h))

    def gallery_exact(self, on: bool = True):
        """keep every row also as float64, as enrolled (frp.h: frp_gallery_exact): the rows behind gallery_distances"""
        self._chk(self._lib.frp_gallery_exact(self._h, 1 if on else 0))

    def gallery_distances(self, q: np.ndarray) -> np.ndarray:
        """Euclidean distances [M, N] (float64) of M queries to the exact rows: face_recognition.face_distance on the device"""
        q = np.asarray(q, dtype=np.float64)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        if q.shape[1] < EMB_DIM:
            q = np.concatenate([q, np.zeros((q.shape[0], EMB_DIM - q.shape[1]))], axis=1)
        q = np.ascontiguousarray(q)
        if q.shape[1] != EMB_DIM:
            raise ValueError("queries must have at most 512 values")
        for _ in range(8):
            n = self.gallery_size()
            out = np.empty((q.shape[0], n), np.float64)
            rc = self._lib.frp_gallery_distances(self._h, _ptr(q), q.shape[0], _ptr(out), n)
            if rc == 0:
                return out
            if self.gallery_size() == n:
                break
        self._chk(rc)
        return out

    def gallery_get_exact(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self.gallery_size() - first if n is None else n
        out = np.empty((n, EMB_DIM), dtype=np.float64)
        self._chk(self._lib.frp_gallery_get_exact(self._h, _ptr(out), first, n))
        return out

    def gallery_remove_row(self, row: int):
        self._chk(self._lib.frp_gallery_remove_row(self._h, row))

    def gallery_size(self) -> int:
        return int(self._lib.frp_gallery_size(self._h))

    def gallery_get(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self.gallery_size() - first if n is None else n
        out = np.empty((n, EMB_DIM), dtype=np.float16)
        self._chk(self._lib.frp_gallery_get(self._h, _ptr(out), first, n))
        return out

    # -- hot path
    @staticmethod
    def _frames(frames: np.ndarray) -> Tuple[np.ndarray, int, int, int, int]:
        if frames.ndim == 3:
            frames = frames[None]
        if frames.ndim != 4 or frames.shape[3] != 3 or frames.dtype != np.uint8:
            raise ValueError("frames must be uint8 [B,H,W,3]")
        frames = np.ascontiguousarray(frames)
        B, H, W, _ = frames.shape
        return frames, B, H, W, W * 3

    @staticmethod
    def _alloc(B, K):
        return dict(boxes=np.zeros((B, K, 4), np.float32), kps=np.zeros((B, K, 5, 2), np.float32),
                    scores=np.zeros((B, K), np.float32), counts=np.zeros((B,), np.int32),
                    emb=np.zeros((B, K, EMB_DIM), np.float32), match_idx=np.full((B, K), -1, np.int32),
                    match_cos=np.full((B, K), -1.0, np.float32))

    def process_frames(self, frames: np.ndarray, max_faces: int = 10, det_thresh: float = 0.5, nms_iou: float = 0.4,
                       flags: int = 0) -> dict:
        frames, B, H, W, rs = self._frames(frames)
        o = self._alloc(B, max_faces)
        self._chk(self._lib.frp_process_frames(self._h, _ptr(frames), B, H, W, rs, max_faces, det_thresh, nms_iou, flags,
                                               _ptr(o["boxes"]), _ptr(o["kps"]), _ptr(o["scores"]), _ptr(o["counts"]),
                                               _ptr(o["emb"]), _ptr(o["match_idx"]), _ptr(o["match_cos"])))
        self._det_batch = B
        return o

    def upload_frames(self, frames: np.ndarray):
        frames, B, H, W, rs = self._frames(frames)
        self._chk(self._lib.frp_upload_frames(self._h, _ptr(frames), B, H, W, rs))
        self._resident = (B, H, W)

    def host_frames(self, B: int, H: int, W: int) -> np.ndarray:
        """page-locked u8 [B,H,W,3] array owned by the engine (freed with it): frames written here can be
        copied to the device while the previous batch is being processed"""
        n = B * H * W * 3
        p = self._lib.frp_host_alloc(self._h, n)
        if not p:
            raise FrpError(self._lib.frp_last_error(self._h).decode())
        buf = (C.c_uint8 * n).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(B, H, W, 3)

    def upload_frames_async(self, frames: np.ndarray):
        """stage the NEXT batch (copy stream); becomes resident at swap_frames()"""
        frames, B, H, W, rs = self._frames(frames)
        self._chk(self._lib.frp_upload_frames_async(self._h, _ptr(frames), B, H, W, rs))
        self._staged = (B, H, W)

    def swap_frames(self):
        self._chk(self._lib.frp_swap_frames(self._h))
        self._resident = self._staged

    def process_resident(self, max_faces: int = 10, det_thresh: float = 0.5, nms_iou: float = 0.4, flags: int = 0):
        self._chk(self._lib.frp_process_resident(self._h, max_faces, det_thresh, nms_iou, flags))
        self._last_k = max_faces

    def synchronize(self):
        self._chk(self._lib.frp_synchronize(self._h))

    def fetch_results(self) -> dict:
        B = self._resident[0]
        o = self._alloc(B, self._last_k)
        self._chk(self._lib.frp_fetch_results(self._h, B, self._last_k, _ptr(o["boxes"]), _ptr(o["kps"]), _ptr(o["scores"]), _ptr(o["counts"]),
                                              _ptr(o["emb"]), _ptr(o["match_idx"]), _ptr(o["match_cos"])))
        return o

    # -- stages
    def detect(self, frames: np.ndarray, max_faces: int = 10, det_thresh: float = 0.5, nms_iou: float = 0.4, flags: int = 0):
        frames, B, H, W, rs = self._frames(frames)
        o = self._alloc(B, max_faces)
        anchor = np.full((B, max_faces), -1, np.int32)
        self._chk(self._lib.frp_detect(self._h, _ptr(frames), B, H, W, rs, max_faces, det_thresh, nms_iou, flags,
                                       _ptr(o["boxes"]), _ptr(o["kps"]), _ptr(o["scores"]), _ptr(o["counts"]), _ptr(anchor)))
        o["anchor_idx"] = anchor
        self._det_batch = B
        return o

    def detect_resident(self, det_hw, max_faces=10, det_thresh=0.5, nms_iou=0.4, flags=0):
        """detect on the resident frames resized to det_hw (pyramid scale); coordinates of the resized image"""
        B = self._resident[0]
        o = self._alloc(B, max_faces)
        anchor = np.full((B, max_faces), -1, np.int32)
        self._chk(self._lib.frp_detect_resident(self._h, B, det_hw[0], det_hw[1], max_faces, det_thresh, nms_iou, flags,
                                                _ptr(o["boxes"]), _ptr(o["kps"]), _ptr(o["scores"]), _ptr(o["counts"]), _ptr(anchor)))
        o["anchor_idx"] = anchor
        self._det_batch = B
        return {k: o[k] for k in ("boxes", "kps", "scores", "counts", "anchor_idx")}

    def det_source(self) -> np.ndarray:
        """the u8 frames the detector last read (resident frames or their pyramid resize)"""
        hs, ws = C.c_int32(), C.c_int32()
        self._chk(self._lib.frp_get_det_source(self._h, None, 0, C.byref(hs), C.byref(ws)))
        out = np.empty((self._resident[0], hs.value, ws.value, 3), np.uint8)
        self._chk(self._lib.frp_get_det_source(self._h, _ptr(out), out.nbytes, C.byref(hs), C.byref(ws)))
        return out

    def finish_faces(self, boxes, kps, scores, counts, max_faces, flags=0):
        """align (from the full-resolution resident frames) + embed + match for caller-supplied landmarks"""
        B = self._resident[0]
        boxes = np.ascontiguousarray(boxes, np.float32).reshape(B, max_faces, 4)
        kps = np.ascontiguousarray(kps, np.float32).reshape(B, max_faces, 5, 2)
        scores = np.ascontiguousarray(scores, np.float32).reshape(B, max_faces)
        counts = np.ascontiguousarray(counts, np.int32).reshape(B)
        o = self._alloc(B, max_faces)
        self._chk(self._lib.frp_finish_faces(self._h, B, _ptr(boxes), _ptr(kps), _ptr(scores), _ptr(counts), max_faces, flags,
                                             _ptr(o["emb"]), _ptr(o["match_idx"]), _ptr(o["match_cos"])))
        o.update(boxes=boxes, kps=kps, scores=scores, counts=counts)
        return o

    def process_frames_pyramid(self, frames, scales=(1.0, 0.5, 0.25), max_faces=10, det_thresh=0.5, nms_iou=0.4,
                               per_scale=64, flags=0):
        """BASELINE config 4: detect at several scales of every frame, merge + NMS across scales
        (pyramid.merge_scales), then align / embed / match from the full-resolution frames."""
        from . import pyramid
        frames, B, H, W, _ = self._frames(frames)
        with self._seq:
            self.upload_frames(frames)
            per = []
            for s in scales:
                hw = pyramid.scaled_size(H, W, s)
                per.append((hw, self.detect_resident(hw, max_faces=per_scale, det_thresh=det_thresh, nms_iou=nms_iou,
                                                     flags=flags & ~FLAG_FORCED_K)))
            boxes, kps, scores, counts = pyramid.merge_scales(per, (H, W), max_faces, nms_iou)
            return self.finish_faces(boxes, kps, scores, counts, max_faces, flags & (FLAG_RGB | FLAG_NO_MATCH))

    def head_maps(self):
        """fp16 head maps [B,H_l,W_l,32] of the last detect/process call, strides 8/16/32."""
        outs = []
        for lv in range(3):
            hl, wl = C.c_int32(), C.c_int32()
            self._chk(self._lib.frp_get_head_map(self._h, lv, None, 0, C.byref(hl), C.byref(wl)))
            a = np.empty((self._det_batch, hl.value, wl.value, 32), dtype=np.float16)
            self._chk(self._lib.frp_get_head_map(self._h, lv, _ptr(a), a.nbytes, C.byref(hl), C.byref(wl)))
            outs.append(a)
        return outs

    def det_prefix(self, n_ops: int) -> np.ndarray:
        """diagnostic: the detector on the RESIDENT frames up to and including op n_ops - 1; that op's output [B,h,w,c] fp16"""
        th, tw, tc = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._lib.frp_debug_det_prefix(self._h, n_ops, None, 0, C.byref(th), C.byref(tw), C.byref(tc)))
        a = np.empty((self._resident[0], th.value, tw.value, tc.value), dtype=np.float16)
        self._chk(self._lib.frp_debug_det_prefix(self._h, n_ops, _ptr(a), a.nbytes, C.byref(th), C.byref(tw), C.byref(tc)))
        return a

    def det_hashes(self, enable: bool = True, fetch: bool = True):
        """diagnostic: per-op output hashes of the last detector pass (uint64[64]); enable keeps them on for later passes"""
        out = np.zeros(64, np.uint64) if fetch else None
        self._chk(self._lib.frp_debug_det_hashes(self._h, 1 if enable else 0, _ptr(out) if fetch else None))
        return out

    def decode_heads(self, heads, canvas_hw, max_faces=10, det_thresh=0.5, nms_iou=0.4, flags=0):
        hs = [np.ascontiguousarray(x, dtype=np.float16) for x in heads]
        B = hs[0].shape[0]
        o = self._alloc(B, max_faces)
        anchor = np.full((B, max_faces), -1, np.int32)
        self._chk(self._lib.frp_decode_heads(self._h, _ptr(hs[0]), _ptr(hs[1]), _ptr(hs[2]), B, canvas_hw[0], canvas_hw[1],
                                             max_faces, det_thresh, nms_iou, flags, _ptr(o["boxes"]), _ptr(o["kps"]),
                                             _ptr(o["scores"]), _ptr(o["counts"]), _ptr(anchor)))
        o["anchor_idx"] = anchor
        return o

    def align(self, frame: np.ndarray, kps: np.ndarray, flags: int = 0) -> np.ndarray:
        frames, _, H, W, rs = self._frames(frame)
        k = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 10)
        out = np.empty((k.shape[0], CHIP, CHIP, 8), dtype=np.float16)
        self._chk(self._lib.frp_align(self._h, _ptr(frames), H, W, rs, _ptr(k), k.shape[0], flags, _ptr(out)))
        return out

    def embed_aligned(self, chips_bgr_u8: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(chips_bgr_u8, dtype=np.uint8).reshape(-1, CHIP, CHIP, 3)
        out = np.empty((c.shape[0], EMB_DIM), dtype=np.float32)
        self._chk(self._lib.frp_embed_aligned(self._h, _ptr(c), c.shape[0], _ptr(out)))
        return out

    def embed_faces(self, frame: np.ndarray, kps: np.ndarray, flags: int = 0) -> np.ndarray:
        frames, _, H, W, rs = self._frames(frame)
        k = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 10)
        out = np.empty((k.shape[0], EMB_DIM), dtype=np.float32)
        self._chk(self._lib.frp_embed_faces(self._h, _ptr(frames), H, W, rs, _ptr(k), k.shape[0], flags, _ptr(out)))
        return out

    def match(self, q: np.ndarray, topk: int = 1):
        """-> (row idx, cosine): [M] for topk == 1, else [M, topk] ordered by (cosine desc, row asc);
        columns beyond the gallery size hold -1 / -2.0"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, EMB_DIM)
        shape = (q.shape[0],) if topk == 1 else (q.shape[0], topk)
        idx = np.empty(shape, np.int32)
        cos = np.empty(shape, np.float32)
        self._chk(self._lib.frp_match(self._h, _ptr(q), q.shape[0], int(topk), _ptr(idx), _ptr(cos)))
        return idx, cos

    def match_scores(self, q: np.ndarray) -> np.ndarray:
        """all cosines [M, N].  The output is sized from gallery_size(); the library re-checks that size under
        its mutex and refuses (nothing written) if a concurrent update changed it -- then size again and retry."""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, EMB_DIM)
        for _ in range(8):
            n = self.gallery_size()
            out = np.empty((q.shape[0], n), np.float32)
            rc = self._lib.frp_match_scores(self._h, _ptr(q), q.shape[0], _ptr(out), n)
            if rc == 0:
                return out
            if self.gallery_size() == n:
                break
        self._chk(rc)
        return out

    def conv2d(self, x, w, bias, stride=1, act=0, slope=None, res=None, flags=0):
        x = np.ascontiguousarray(x, dtype=np.float16)
        w = np.ascontiguousarray(w, dtype=np.float16)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        N, H, W, Cin = x.shape
        Cout, k = w.shape[0], w.shape[1]
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = np.empty((N, Ho, Wo, Cout), dtype=np.float32 if (flags & 2) else np.float16)
        rh = rw = 0
        if res is not None:
            res = np.ascontiguousarray(res, dtype=np.float16)
            rh, rw = res.shape[1], res.shape[2]
        if slope is not None:
            slope = np.ascontiguousarray(slope, dtype=np.float32)
        self._chk(self._lib.frp_conv2d_nhwc(self._h, _ptr(x), N, H, W, Cin, _ptr(w), Cout, k, stride, _ptr(bias), _ptr(slope),
                                            _ptr(res), rh, rw, act, flags, _ptr(out)))
        return out

    def conv2d_f8(self, x8, w8, wscale, bias, act=0, slope=None, res=None, flags=0, in_scale=1.0, out_scale=1.0,
                  out_fp8=False, copy_fp8=False):
        """3x3 stride-1 conv on E4M3 operands (uint8 codes).  -> fp16 output (or uint8 codes with out_fp8), plus the
        uint8 copy with copy_fp8"""
        x8 = np.ascontiguousarray(x8, dtype=np.uint8)
        w8 = np.ascontiguousarray(w8, dtype=np.uint8)
        N, H, W, Cin = x8.shape
        Cout = w8.shape[0]
        wscale = np.ascontiguousarray(wscale, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        if slope is not None:
            slope = np.ascontiguousarray(slope, dtype=np.float32)
        if res is not None:
            res = np.ascontiguousarray(res, dtype=np.float16)
        out = np.empty((N, H, W, Cout), np.uint8 if out_fp8 else np.float16)
        out2 = np.empty((N, H, W, Cout), np.uint8) if copy_fp8 else None
        self._chk(self._lib.frp_conv2d_f8(self._h, _ptr(x8), N, H, W, Cin, _ptr(w8), Cout, _ptr(wscale), _ptr(bias), _ptr(slope),
                                          _ptr(res), act, flags | (64 if out_fp8 else 0), in_scale, out_scale, _ptr(out), _ptr(out2)))
        return (out, out2) if copy_fp8 else out

    def _lab(self, name: str):
        if not hasattr(self._lib, name):
            raise RuntimeError(f"{name} lives in the lab build only: `make -C face-recognition-platform_amd/csrc lab` and run with "
                               "FRP_LIB=face-recognition-platform_amd/libfrp_lab.so (tools/ do that through tools/_lab.py)")
        return getattr(self._lib, name)

    def conv_bench(self, N, H, W, Cin, Cout, k=3, stride=1, act=0, flags=0, with_res=False, iters=20, stamps=False):
        self._lab("frp_conv_bench")
        ms = C.c_float()
        st = np.zeros((256, 8), np.uint64) if stamps else None
        self._chk(self._lib.frp_conv_bench(self._h, N, H, W, Cin, Cout, k, stride, act, flags, int(with_res), iters, C.byref(ms), _ptr(st)))
        return (float(ms.value), st) if stamps else float(ms.value)

    def mfma_peak(self, waves_per_simd=1, iters=20000) -> float:
        t = C.c_float()
        self._lab("frp_mfma_peak")
        self._chk(self._lib.frp_mfma_peak(self._h, waves_per_simd, iters, C.byref(t)))
        return float(t.value)

    def mfma_lds_peak(self, reads_per_4_mfma: int = 4, iters: int = 5000) -> float:
        """TFLOP/s of the conv k-step's instruction mix alone (8 waves per CU, 64x64 wave tiles, fragments from LDS
        by ds_read_b128, random data, no DMA / barriers / epilogue): the ceiling of that wave layout"""
        return self.mfma_peak(16 * int(reads_per_4_mfma) + 2, iters)

    def kstep_lab(self, variant: int, iters: int = 3000) -> float:
        """TFLOP/s of the conv k-step's inner loop in isolation under schedule `variant` (csrc/kstep_lab.hip)"""
        t = C.c_float()
        self._lab("frp_kstep_lab")
        self._chk(self._lib.frp_kstep_lab(self._h, int(variant), int(iters), C.byref(t)))
        return float(t.value)

    def counters(self) -> dict:
        c = FrpCounters()
        self._chk(self._lib.frp_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def set_profile(self, on: bool):
        """per-stage HIP-event timing on / off (on: every process call ends in a stream synchronise)"""
        self._chk(self._lib.frp_set_profile(self._h, int(bool(on))))

    def reset_counters(self):
        self._chk(self._lib.frp_reset_counters(self._h))
