"""Image ingest for the REST upload paths: encoded stills -> device frames.

Baseline JPEG uploads (what phone cameras and browsers send; routes/face.py:177-185 receives them as bytes) are decoded ON
THE WAY to the device (round 4): their bit streams - the serial part - on host threads inside `frp_upload_jpeg_async`, the
dequantisation, inverse DCT, chroma upsampling and colour conversion by HIP kernels on the engine's copy stream, straight into
the staging frame buffer, bit for bit what PIL would have decoded (`device_jpeg_batch` decides per batch; anything else - PNG,
progressive JPEG, mixed sizes, a short last batch - takes the host path below).

The reference decodes every upload with PIL (`face_recognition.load_image_file`, routes/face.py:216,404,976) into a
pageable array, which a naive port would hand to hipMemcpy: the runtime then bounces it through its own staging buffer
before the DMA.  Here the decoder's output goes into PAGE-LOCKED staging owned by the engine (`frp_host_alloc`) and that
same memory is the source of the asynchronous H2D copy on the engine's copy stream (`frp_upload_frames_async`): one
host copy (decoder buffer -> pinned; PIL cannot decode into caller memory), no runtime bounce, and - two staging
buffers - the decode + upload of batch t+1 runs while the GPU processes batch t.

Video (RTSP / H.264) ingest is NOT covered: the image has no codec library or hardware-decode API.
"""
from __future__ import annotations

import io
from typing import Iterable, Iterator, List, Sequence, Tuple, Union

import numpy as np

from . import native

Source = Union[str, bytes, np.ndarray]


def decode_rgb(src: Source, hw: Tuple[int, int]) -> np.ndarray:
    """one still (path, encoded bytes, or an already decoded RGB array) -> u8 RGB [H,W,3]; raises on a size mismatch"""
    if isinstance(src, np.ndarray):
        img = src
    else:
        from PIL import Image
        with Image.open(io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src) as im:
            img = np.asarray(im.convert("RGB"))
    if img.shape != (hw[0], hw[1], 3) or img.dtype != np.uint8:
        raise ValueError(f"image is {img.shape} {img.dtype}, the staging buffer holds {hw[0]}x{hw[1]}x3 uint8")
    return img


def device_jpeg_batch(sources: Sequence[Source], batch: int, hw: Tuple[int, int]):
    """-> the batch as a list of JPEG byte strings when the device decoder covers ALL of it (a full batch of baseline JPEG
    stills of the staging buffer's size and one chroma sampling), else None"""
    if len(sources) != batch:
        return None
    out, sampling = [], None
    for s in sources:
        if isinstance(s, np.ndarray):
            return None
        if isinstance(s, str):
            with open(s, "rb") as f:
                s = f.read()
        if not isinstance(s, bytes):               # (a bytes subclass - mjpeg.JpegFrame - is passed on as it is: no copy)
            s = bytes(s)
        info = native.jpeg_info(s)
        if info is None or (info["height"], info["width"]) != tuple(hw):
            return None
        key = (info["components"], info["h_samp"][0], info["v_samp"][0])
        if sampling is not None and key != sampling:
            return None
        sampling = key
        out.append(s)
    return out


def batched(sources: Sequence[Source], batch: int) -> Iterator[Sequence[Source]]:
    """an upload of any length as consecutive batches of at most `batch` stills (what StagedIngest.run consumes)"""
    for i in range(0, len(sources), batch):
        yield sources[i:i + batch]


class StagedIngest:
    """Two page-locked staging buffers of B x H x W RGB frames on one engine."""

    def __init__(self, engine, batch: int, height: int, width: int, device_jpeg: bool = True):
        self.eng, self.B, self.H, self.W = engine, batch, height, width
        self._stage = [engine.host_frames(batch, height, width) for _ in range(2)]
        self.device_jpeg = device_jpeg and hasattr(engine, "upload_jpeg_async")
        self.device_decoded = 0            # batches that took the device decoder

    def _stage_batch(self, slot: int, sources: Sequence[Source]) -> Tuple[int, int]:
        """bring one batch into the engine's staging frame buffer -> (real frames, process flags for it): JPEG batches are
        decoded on the way (BGR frames, no flag), everything else by PIL into page-locked staging (RGB frames)"""
        jp = device_jpeg_batch(sources, self.B, (self.H, self.W)) if self.device_jpeg else None
        if jp is not None:
            self.eng.upload_jpeg_async(jp)
            self.device_decoded += 1
            return len(jp), 0
        n = self.decode_into(slot, sources)
        self.eng.upload_frames_async(self._stage[slot])
        return n, native.FLAG_RGB

    def decode_into(self, slot: int, sources: Sequence[Source]) -> int:
        """decode up to B stills into staging buffer `slot`; frames beyond len(sources) are zeroed.  -> count.
        More than B sources is an error (nothing is dropped silently): chunk the upload with `batched()`."""
        buf = self._stage[slot]
        n = len(sources)
        if n > self.B:
            raise ValueError(f"{n} images for a staging buffer of {self.B}: split the upload into batches (ingest.batched)")
        for i in range(n):
            np.copyto(buf[i], decode_rgb(sources[i], (self.H, self.W)))
        if n < self.B:
            buf[n:] = 0
        return n

    def run(self, batches: Iterable[Sequence[Source]], max_faces: int = 10, det_thresh: float = 0.5, nms_iou: float = 0.4,
            flags: int = 0) -> Iterator[Tuple[int, dict]]:
        """detect + embed + match every batch of stills; yields (number of real frames, result dict) per batch.
        Decode and upload of batch t+1 overlap the GPU work of batch t."""
        eng = self.eng
        it = iter(batches)
        with eng.sequence():
            first = next(it, None)
            if first is None:
                return
            n_cur, fl_cur = self._stage_batch(0, first)
            eng.swap_frames()
            slot = 1
            while True:
                nxt = next(it, None)
                eng.process_resident(max_faces, det_thresh=det_thresh, nms_iou=nms_iou, flags=flags | fl_cur)   # asynchronous
                n_next = fl_next = 0
                if nxt is not None:
                    n_next, fl_next = self._stage_batch(slot, nxt)   # host decode / entropy decode while the GPU works; the copy
                out = eng.fetch_results()                            # (and the device half of a JPEG decode) overlap the running pass
                yield n_cur, out
                if nxt is None:
                    return
                eng.swap_frames()
                n_cur, fl_cur, slot = n_next, fl_next, slot ^ 1
