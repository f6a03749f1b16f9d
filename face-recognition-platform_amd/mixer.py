"""Stream mixer (BASELINE config 5: "16 synthetic RTSP streams @720p mixed batch", 2 streams per GPU).

The reference reads ONE frame per camera per poll: `cap.read()` `frame_skip` times keeping the last
(backend/app/routes/camera.py:204-213), an optional per-camera `fps_limit` sleep (:216-221), captures opened and
re-opened through cv2.VideoCapture (backend/app/state.py:348-450).  Here the streams assigned to this GPU
(dist.streams_of_rank) are MIXED into device batches: slot i of a batch takes the next kept frame of stream
i mod n_streams, so a batch of B frames carries B / n_streams consecutive kept frames of every live stream, interleaved
frame by frame.  Frames are written - decoded, when the stream delivers encoded stills (JPEG / PNG bytes, what an MJPEG
or snapshot camera sends) - straight into page-locked batch buffers by a small thread pool, so that the upload of a batch
is a DMA at PCIe rate and overlaps the device work of the batches in flight (FaceService.process_stream / lanes.py).

The image has no video codec (no cv2, no ffmpeg, no hardware-decode API): H.264 / RTSP de-packetisation is out of reach
offline.  A stream is therefore anything with the capture interface the reference uses - `isOpened()`, `read() ->
(ok, frame)` - whose frames are BGR u8 arrays or encoded stills; `SyntheticStream` is the seeded stand-in used by the
tests and by bench.py --workload config5.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np


class SyntheticStream:
    """Capture stand-in: `n_frames` frames (a [n, H, W, 3] BGR array, or a list of encoded stills) served by read() in
    order; `fail_at` makes one read fail (a dropped RTSP connection), after which the stream is closed until open()."""

    def __init__(self, frames, fail_at: Optional[int] = None):
        self._frames = frames
        self._i = 0
        self._open = True
        self._fail_at = fail_at
        self.reads = 0

    def isOpened(self) -> bool:
        return self._open

    def open(self, source=None) -> bool:
        self._open = True
        return True

    def read(self):
        self.reads += 1
        if not self._open or self._i >= len(self._frames):
            return False, None
        if self._fail_at is not None and self._i == self._fail_at:
            self._fail_at = None
            self._open = False
            return False, None
        f = self._frames[self._i]
        self._i += 1
        return True, f


def _decode_into(dst: np.ndarray, item) -> None:
    """one frame into its batch slot: arrays are copied, encoded stills (bytes) decoded by PIL (RGB) and stored as BGR"""
    if isinstance(item, (bytes, bytearray, memoryview)):
        import io
        from PIL import Image
        with Image.open(io.BytesIO(bytes(item))) as im:
            rgb = np.asarray(im.convert("RGB"))
        if rgb.shape != dst.shape:
            raise ValueError(f"stream delivered {rgb.shape}, the batch holds {dst.shape}")
        np.copyto(dst, rgb[..., ::-1])
    else:
        a = np.asarray(item)
        if a.shape != dst.shape or a.dtype != np.uint8:
            raise ValueError(f"stream delivered {a.shape} {a.dtype}, the batch holds {dst.shape} uint8")
        np.copyto(dst, a)


class StreamMixer:
    """Mixes the kept frames of several streams into batches of `batch` frames.

    streams      : {stream id: capture-like object}; ids are served round-robin in the order given
    buffers      : page-locked [batch, H, W, 3] u8 arrays used in rotation (FaceService.frame_buffer /
                   Engine.host_frames); at least 3 x lanes + 2 of them, so that a buffer is not rewritten while its batch
                   is still in flight (process_stream's pipeline takes 3 x lanes batches ahead).  Plain numpy arrays work too (tests).
    frame_skip   : reads per kept frame, the last one is kept (camera.py:204-213); a failed read ends the stream's turn
                   (after ONE reopen attempt for a closed capture, camera.py:185-200)
    fps_limit    : {stream id: fps} or a number for all: minimum spacing of kept frames (camera.py:216-221)
    encoded      : the streams deliver ENCODED frames (mjpeg.MjpegCapture: JPEG bytes) and the batches stay encoded - the mixer
                   yields (mjpeg.JpegBatch, meta) and the engine decodes a batch on the way to the device
                   (frp_upload_jpeg_async); `buffers` then only gives the batch geometry (their [1:3] shape) and may be plain
                   arrays.  A slot whose stream has ended repeats another slot's frame (its meta stays None: the result is dropped).
    Yields (buffer, meta) with meta[i] = (stream id, index of the kept frame in that stream) or None for a slot left
    empty (zeroed) because its stream had ended; stops when no stream delivers any more.
    """

    def __init__(self, streams: Dict[Any, Any], batch: int, buffers: Sequence[np.ndarray], frame_skip: int = 1,
                 fps_limit=None, decode_workers: int = 4, clock: Callable[[], float] = time.time,
                 sleep: Callable[[float], None] = time.sleep, encoded: bool = False):
        if not streams or batch < 1 or not buffers:
            raise ValueError("need at least one stream, one batch slot and one buffer")
        for b in buffers:
            if b.ndim != 4 or b.shape[0] != batch or b.shape[3] != 3 or b.dtype != np.uint8:
                raise ValueError("buffers must be uint8 [batch, H, W, 3]")
        self.streams = dict(streams)
        self.order = list(self.streams.keys())
        self.batch = batch
        self.buffers = list(buffers)
        self.frame_skip = max(1, int(frame_skip))
        self.fps = ({k: fps_limit for k in self.order} if isinstance(fps_limit, (int, float)) else dict(fps_limit or {}))
        self._clock, self._sleep = clock, sleep
        self._kept = {k: 0 for k in self.order}
        self._last_t: Dict[Any, float] = {}
        self._ended = set()
        self._pool = ThreadPoolExecutor(max_workers=max(1, decode_workers))
        self._lock = threading.Lock()
        self.dropped_reads = 0
        self.encoded = bool(encoded)

    def close(self) -> None:
        self._pool.shutdown(wait=True)

    # one kept frame of a stream, or None when it has ended / failed (camera.py:176-213 restated per stream)
    def _next_kept(self, sid):
        if sid in self._ended:
            return None
        cap = self.streams[sid]
        if cap is None:
            self._ended.add(sid)
            return None
        if not cap.isOpened():
            try:
                cap.open(sid)
            except Exception:
                pass
            if not cap.isOpened():
                self._ended.add(sid)
                return None
        lim = self.fps.get(sid)
        if lim:
            now = self._clock()
            wait = self._last_t.get(sid, -1e30) + 1.0 / float(lim) - now
            if wait > 0:
                self._sleep(wait)
        frame = None
        for _ in range(self.frame_skip):
            ok, cand = cap.read()
            if not ok:
                if cap.isOpened():          # end of stream (a closed capture gets its reopen attempt next turn)
                    self._ended.add(sid)
                self.dropped_reads += 1
                return None
            frame = cand
        self._last_t[sid] = self._clock()
        idx = self._kept[sid]
        self._kept[sid] += 1
        return idx, frame

    def _iter_encoded(self):
        from .mjpeg import JpegBatch
        n = len(self.order)
        hw = self.buffers[0].shape[1:3]
        while True:
            meta: List[Optional[Tuple[Any, int]]] = [None] * self.batch
            frames: List[Any] = [None] * self.batch
            for slot in range(self.batch):
                got = self._next_kept(self.order[slot % n])
                if got is None:
                    continue
                idx, frame = got
                if not isinstance(frame, (bytes, bytearray)):
                    raise ValueError("StreamMixer(encoded=True): the streams must deliver encoded frames (bytes)")
                meta[slot] = (self.order[slot % n], idx)
                frames[slot] = frame
            filler = next((f for f in frames if f is not None), None)
            if filler is None:
                return
            yield JpegBatch([f if f is not None else filler for f in frames], hw), meta

    def __iter__(self) -> Iterator[Tuple[np.ndarray, List[Optional[Tuple[Any, int]]]]]:
        if self.encoded:
            yield from self._iter_encoded()
            return
        n = len(self.order)
        bi = 0
        while True:
            buf = self.buffers[bi % len(self.buffers)]
            meta: List[Optional[Tuple[Any, int]]] = [None] * self.batch
            jobs = []
            for slot in range(self.batch):
                sid = self.order[slot % n]
                got = self._next_kept(sid)
                if got is None:
                    continue
                idx, frame = got
                meta[slot] = (sid, idx)
                jobs.append(self._pool.submit(_decode_into, buf[slot], frame))
            if not jobs:
                return
            for slot in range(self.batch):
                if meta[slot] is None:
                    buf[slot] = 0
            for j in jobs:
                j.result()                  # decode errors surface here
            bi += 1
            yield buf, meta


def demix(meta: List[Optional[Tuple[Any, int]]], per_frame_results: List[Any]) -> Dict[Any, List[Tuple[int, Any]]]:
    """per-batch results (one entry per slot, e.g. FaceService.process_frames output) back to {stream id: [(frame index,
    result), ...]} in stream order; empty slots are dropped"""
    out: Dict[Any, List[Tuple[int, Any]]] = {}
    for m, r in zip(meta, per_frame_results):
        if m is not None:
            out.setdefault(m[0], []).append((m[1], r))
    return out


def run_mixed(service, mixer: StreamMixer, **kw) -> Iterator[Dict[Any, List[Tuple[int, Any]]]]:
    """mixer -> FaceService.process_stream (two batches in flight) -> per batch {stream id: [(frame index, faces), ...]}"""
    import collections
    metas = collections.deque()

    def batches():
        for buf, meta in mixer:
            metas.append(meta)
            yield buf
    for res in service.process_stream(batches(), **kw):
        yield demix(metas.popleft(), res)
