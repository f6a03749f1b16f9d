"""Image ingest for the REST upload paths (first step): encoded stills -> device frames.

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


def batched(sources: Sequence[Source], batch: int) -> Iterator[Sequence[Source]]:
    """an upload of any length as consecutive batches of at most `batch` stills (what StagedIngest.run consumes)"""
    for i in range(0, len(sources), batch):
        yield sources[i:i + batch]


class StagedIngest:
    """Two page-locked staging buffers of B x H x W RGB frames on one engine."""

    def __init__(self, engine, batch: int, height: int, width: int):
        self.eng, self.B, self.H, self.W = engine, batch, height, width
        self._stage = [engine.host_frames(batch, height, width) for _ in range(2)]

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
        flags |= native.FLAG_RGB
        it = iter(batches)
        with eng.sequence():
            first = next(it, None)
            if first is None:
                return
            n_cur = self.decode_into(0, first)
            eng.upload_frames_async(self._stage[0])
            eng.swap_frames()
            slot = 1
            while True:
                nxt = next(it, None)
                eng.process_resident(max_faces, det_thresh=det_thresh, nms_iou=nms_iou, flags=flags)   # asynchronous
                n_next = 0
                if nxt is not None:
                    n_next = self.decode_into(slot, nxt)          # host decode while the GPU works
                    eng.upload_frames_async(self._stage[slot])    # copy stream: overlaps the running pass
                out = eng.fetch_results()                         # waits for the pass
                yield n_cur, out
                if nxt is None:
                    return
                eng.swap_frames()
                n_cur, slot = n_next, slot ^ 1
