"""Two batches in flight on one GPU.

Every conv launch of the hot path is a persistent kernel of one workgroup per compute unit; its last round of tiles
leaves CUs idle (490 tiles on 256 CUs), its workgroups finish a few microseconds apart, and the next launch of the
same stream cannot start before the last of them has.  A second, independent batch on ANOTHER stream fills exactly
those holes: its workgroups are dispatched onto compute units as the first stream's kernel drains.  Measured on
MI355X (tools/dual_probe.py, bench.py --lanes 2; 32 x 1080p, 10 faces per frame, IResNet-100): 13.2-13.6 ms per batch
with two batches in flight against 14.2-14.5 ms with one (+6...8 % faces/s); a third lane gains nothing.

A lane is a complete engine - one C-ABI handle (include/frp.h): private stream, activation buffers, its own copy of
the weights and of the gallery - so nothing is shared between the batches in flight and every batch gets bit for bit
the result a single engine would give it.  Each lane is driven by its own host thread (ctypes releases the GIL inside
the C calls): the hardware scheduler does not serve two queues evenly, and a single thread that waits for the lanes in
a fixed order leaves the favoured lane idle while it waits for the other one (measured: slower than one lane).
The caller's loop (reference: one pass per poll, routes/camera.py:225-259) only changes in WHEN it reads a result:
results are handed back in submission order, at most `n_lanes` batches later.
"""
from __future__ import annotations

import threading
from typing import Any, Callable, Dict, Iterable, Iterator, List, Sequence

import numpy as np

from . import native


class Deferred:
    """What a prefetch-mode worker may return instead of a result: `finish()` builds the result from what the device handed
    back (host-only work: no device call, no lock of the pipeline).  run_ordered runs it where the lane's thread would
    otherwise sit waiting - inside the worker's NEXT call, after that call has put its batch on the device
    (`take_next.idle()`), or at once when the worker has no next item in hand."""
    __slots__ = ("finish",)

    def __init__(self, finish: Callable[[], Any]):
        self.finish = finish


def run_ordered(batches: Iterable[Any], workers: Sequence[Callable[..., Any]], prefetch: bool = False) -> Iterator[Any]:
    """The one in-order, bounded-look-ahead pipeline behind Lanes.run and FaceService.process_stream: `workers[i]` (one
    host thread each) turn items of `batches` into results; results are yielded in submission order; at most
    2 x len(workers) items are taken ahead of the consumer.  An exception of the iterator or of a worker is raised in
    the consumer.  When the consumer abandons the generator (close(), break, an exception in its loop body) every
    worker - also one parked waiting for an item - leaves after the item it is working on.

    The source is read by a FEEDER thread of its own, never by a worker: `next(batches)` may block for as long as the
    source likes (a live camera sleeps for its fps limit, waits in cap.read and decodes: mixer.StreamMixer) and may itself
    call into the service (a generator that enrols or looks up identities) - it runs outside every lock of this
    pipeline and of the caller, and what it delivers waits in a queue of at most 2 x len(workers) items.

    `prefetch=True`: workers are called as fn(item, take_next); `take_next()` - callable once per item, NEVER blocking:
    it only takes an item the feeder has already delivered - claims the item this worker will get NEXT (or None:
    nothing ready, source dry) so that the worker can start moving it to the device while the current item is being
    processed (FaceService.process_stream: the upload of batch t+1 overlaps the kernels of batch t on the same lane).
    A worker may therefore call it while holding locks of its own (the gallery's shared lock).
    A prefetch-mode worker may return a `Deferred`: its `finish()` (host-side result building) then runs on the same thread
    under the device work of that worker's next item - the worker calls `take_next.idle()` once its next batch is on the
    device - or immediately if the worker has not claimed a next item (so a result is never held back by a dry source).
    With deferral a lane holds three items (one being finished, one on the device, one staged), so the look-ahead bound
    is 3 x len(workers) in prefetch mode, 2 x len(workers) otherwise."""
    it = iter(batches)
    n = len(workers)
    if n < 1:
        raise ValueError("run_ordered needs at least one worker")
    cv = threading.Condition()
    st = {"next": 0, "yielded": 0, "fed": 0, "done": {}, "ready": [], "src_done": False, "stopped": False, "error": None}

    depth = (3 if prefetch else 2) * n

    def feeder() -> None:
        while True:
            with cv:
                while st["fed"] - st["yielded"] >= depth and not st["stopped"] and st["error"] is None:
                    cv.wait()
                if st["stopped"] or st["error"] is not None:
                    return
            try:
                item = next(it)                    # no lock held: the source may block, sleep or call back into the caller
            except StopIteration:
                with cv:
                    st["src_done"] = True
                    cv.notify_all()
                return
            except BaseException as ex:            # the caller's iterator failed: surface it in the consumer
                with cv:
                    st["error"] = ex
                    cv.notify_all()
                return
            with cv:
                st["ready"].append(item)
                st["fed"] += 1
                cv.notify_all()

    def claim(block: bool):
        """(index, item) of the next delivered item, or None; call with cv held.  block=False returns at once."""
        while block and not st["ready"] and not st["src_done"] and not st["stopped"] and st["error"] is None:
            cv.wait()
        if st["stopped"] or st["error"] is not None or not st["ready"]:
            return None
        item = st["ready"].pop(0)
        t = st["next"]
        st["next"] += 1
        return t, item

    def loop(fn: Callable[..., Any]) -> None:
        ahead = None                     # the item this worker claimed early (prefetch)
        pending = None                   # (index, Deferred) of this worker's previous item

        def publish(t: int, out: Any) -> None:
            with cv:
                st["done"][t] = out
                cv.notify_all()

        def idle() -> None:
            nonlocal pending
            if pending is not None:
                (t_prev, d), pending = pending, None
                publish(t_prev, d.finish())

        while True:
            if ahead is not None:
                (t, item), ahead = ahead, None
            else:
                with cv:
                    got = claim(True)
                if got is None:
                    return
                t, item = got
            try:
                if prefetch:
                    def take_next():
                        nonlocal ahead
                        if ahead is None:
                            with cv:
                                ahead = claim(False)
                        return None if ahead is None else ahead[1]
                    take_next.idle = idle
                    out = fn(item, take_next)
                    idle()                                   # a worker that never called it: before this item's own result
                    if isinstance(out, Deferred):
                        if ahead is not None:
                            pending = (t, out)               # finished inside the call for `ahead`
                            continue
                        out = out.finish()
                else:
                    out = fn(item)
            except BaseException as ex:
                with cv:
                    st["error"] = ex
                    cv.notify_all()
                return
            publish(t, out)

    feed = threading.Thread(target=feeder, daemon=True)
    threads = [threading.Thread(target=loop, args=(fn,), daemon=True) for fn in workers]
    feed.start()
    for th in threads:
        th.start()
    try:
        while True:
            with cv:
                while (st["yielded"] not in st["done"] and st["error"] is None
                       and not (st["src_done"] and not st["ready"] and st["yielded"] >= st["next"])):
                    cv.wait()
                if st["error"] is not None:
                    raise st["error"]
                if st["yielded"] not in st["done"]:
                    return
                out = st["done"].pop(st["yielded"])
                st["yielded"] += 1
                cv.notify_all()
            yield out
    finally:
        with cv:
            st["stopped"] = True         # consumer gone (or done): workers stop after their current item
            cv.notify_all()
        for th in threads:
            th.join()
        # The feeder may be parked inside the source's `next()` (a live camera that has no frame yet): it is a daemon thread and
        # leaves with the source's next item or its end.  Until then the SOURCE ITERATOR IS STILL IN USE by that thread - a caller
        # that abandons the results early must not reuse or close a generator source before it has produced once more
        # ("generator already executing"); sources with a close() of their own (MjpegCapture.release, StreamMixer.close) end it.
        feed.join(timeout=0.2)


class Lanes:
    def __init__(self, device: int = 0, n_lanes: int = 2, **engine_kwargs):
        if n_lanes < 1:
            raise ValueError("n_lanes must be >= 1")
        self.engines: List[native.Engine] = [native.Engine(device, **engine_kwargs) for _ in range(n_lanes)]

    # ---- state that every lane needs a copy of
    def load_weights(self, blob: bytes) -> None:
        for e in self.engines:
            e.load_weights(blob)

    def gallery_set(self, rows: np.ndarray) -> None:
        for e in self.engines:
            e.gallery_set(rows)

    def gallery_set_device(self, dev_ptr: int, n: int) -> None:
        for e in self.engines:
            e.gallery_set_device(dev_ptr, n)

    def gallery_size(self) -> int:
        return self.engines[0].gallery_size()

    def close(self) -> None:
        for e in self.engines:
            e.close()

    def run(self, batches: Iterable[np.ndarray], max_faces: int = 10, det_thresh: float = 0.5, nms_iou: float = 0.4,
            flags: int = 0) -> Iterator[Dict[str, np.ndarray]]:
        """detect + embed + match every batch of host frames [B,H,W,3]; yields the result dicts in submission order.
        Batch t+1 is pulled from `batches` and runs on another lane while batch t is on the device; at most
        2 x n_lanes batches are taken ahead of the consumer."""
        def on(e: native.Engine):
            return lambda frames: e.process_frames(frames, max_faces=max_faces, det_thresh=det_thresh, nms_iou=nms_iou, flags=flags)
        return run_ordered(batches, [on(e) for e in self.engines])
