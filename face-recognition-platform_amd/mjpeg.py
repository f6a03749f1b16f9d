"""Motion-JPEG ingest: camera streams whose frames are baseline JPEG stills - the one video format that needs
no codec library, and the one USB cameras ("MJPG") and most IP cameras (`multipart/x-mixed-replace` over HTTP) deliver.

The reference reads cameras through `cv2.VideoCapture(source)` / `cap.read()` (backend/app/routes/camera.py:185-221,
backend/app/state.py:348-450), which demuxes and decodes on the host.  Here the container is taken apart on the host (this
module: byte streams of back-to-back JPEGs, HTTP multipart bodies, RIFF/AVI files with MJPG video chunks), the frames stay
ENCODED until they reach the engine, and `frp_upload_jpeg_async` decodes a whole batch on the way to the device (bit streams on
host threads, pixels on the GPU's copy stream; csrc/jpeg_host.cpp, csrc/jpeg_kernels.hip) - bit for bit what PIL / libjpeg
would have produced, including frames without a DHT segment (the Annex K tables are then implied, as in libjpeg-turbo).

`MjpegCapture` has the cv2.VideoCapture surface the reference's loop uses (`isOpened / read / open / release`), so it can stand
where a capture object stands (mixer.StreamMixer, camera_loop.process_camera_sync); `read()` hands out `JpegFrame` - encoded
bytes - which `mixer.StreamMixer(encoded=True)` batches as `JpegBatch` and `FaceService.process_stream / process_frames`
accept in place of a pixel array.  H.264 / H.265 / RTSP stay out of reach offline (no codec library, no hardware-decode API
in the image).
"""
from __future__ import annotations

import io
import struct
from typing import Any, BinaryIO, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np


class JpegFrame(bytes):
    """one encoded video frame (a complete JPEG, SOI ... EOI)"""
    __slots__ = ()


class JpegBatch(list):
    """a batch of encoded frames of ONE geometry (what StreamMixer(encoded=True) yields and FaceService accepts);
    `hw` = (height, width) of every frame"""

    def __init__(self, frames: Iterable[bytes], hw: Tuple[int, int]):
        super().__init__(frames)
        self.hw = (int(hw[0]), int(hw[1]))

    @property
    def shape(self) -> Tuple[int, int, int, int]:
        return (len(self), self.hw[0], self.hw[1], 3)

    def decode(self) -> np.ndarray:
        """host decode (PIL) -> u8 BGR [B,H,W,3]: the path for engines without the device decoder and for batches it does
        not cover (progressive frames, mixed chroma sampling)"""
        from PIL import Image
        out = np.empty(self.shape, np.uint8)
        for i, j in enumerate(self):
            with Image.open(io.BytesIO(j)) as im:
                rgb = np.asarray(im.convert("RGB"))
            if rgb.shape != out.shape[1:]:
                raise ValueError(f"frame {i} is {rgb.shape}, the batch holds {out.shape[1:]}")
            out[i] = rgb[..., ::-1]
        return out


# Sizes taken from a stream (Content-Length, AVI chunk sizes) and frames that never end are bounded by this: a larger frame is
# dropped and the demuxer looks for the next start-of-image (a 4K 4:4:4 JPEG at quality 100 is ~25 MB).
MAX_FRAME_BYTES = 32 << 20

# ---------------------------------------------------------------------------------------------- finding frames in a byte stream

def jpeg_end(buf: Union[bytes, bytearray], start: int = 0, state: Optional[list] = None) -> Optional[int]:
    """`buf[start:]` begins with SOI: -> index one past this JPEG's EOI, or None while the frame is still incomplete.
    `state` (a list owned by the caller, [] for a new frame): where the walk stopped, so that the next call - same frame, more
    bytes - continues there instead of walking the frame from its start again (a 25 MB frame arriving in 64 KiB chunks).
    Walks the marker segments by their lengths (an EXIF thumbnail inside APP1 carries its own SOI / EOI and must not end the
    frame) and scans only entropy-coded data for the next marker (0xFF followed by anything but 0x00 / RSTn)."""
    n = len(buf)
    if n - start < 4:
        return None
    if buf[start] != 0xFF or buf[start + 1] != 0xD8:
        raise ValueError("not at a JPEG start-of-image marker")
    i, in_scan = (state[0], state[1]) if state else (start + 2, False)

    def parked(pos, scanning):
        if state is not None:
            state[:] = [pos, scanning]
        return None
    while True:
        if in_scan:                                # (resumed inside entropy-coded data)
            in_scan = False
            while True:
                j = buf.find(b"\xff", i)
                if j < 0 or j + 1 >= n:
                    return parked(max(i, n - 1) if j < 0 else j, True)
                nxt = buf[j + 1]
                if nxt == 0x00 or 0xD0 <= nxt <= 0xD7:
                    i = j + 2
                elif nxt == 0xFF:
                    i = j + 1
                else:
                    i = j
                    break
            continue
        if i + 2 > n:
            return parked(i, False)
        if buf[i] != 0xFF:
            raise ValueError("corrupt JPEG: marker expected")
        i0 = i
        while i < n and buf[i] == 0xFF:            # fill bytes
            i += 1
        if i >= n:
            return parked(i0, False)
        m = buf[i]
        i += 1
        if m == 0xD9:
            return i
        if m == 0xD8:
            raise ValueError("corrupt JPEG: a new start-of-image before the end of this frame")      # a frame cut short
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        if i + 2 > n:
            return parked(i0, False)
        seg = (buf[i] << 8) | buf[i + 1]
        if seg < 2:
            raise ValueError("corrupt JPEG: segment length")
        if i + seg > n:
            return parked(i0, False)
        i += seg
        if m == 0xDA:                              # entropy-coded data up to the next real marker
            in_scan = True


def split_stream(chunks: Iterable[bytes], max_frame_bytes: int = MAX_FRAME_BYTES) -> Iterator[JpegFrame]:
    """byte chunks of back-to-back JPEG frames (a raw .mjpeg file, a camera pipe; bytes between frames - multipart headers,
    padding - are skipped) -> complete frames.  A damaged frame - or one that has not ended after `max_frame_bytes` - is
    dropped at the next start-of-image; the buffer never holds more than one such frame plus a chunk."""
    buf = bytearray()
    state: list = []                               # where jpeg_end stopped in the frame at buf[0]
    for chunk in chunks:
        if not chunk:
            continue
        buf += chunk
        while True:
            if not state:
                s = buf.find(b"\xff\xd8\xff")
                if s < 0:
                    del buf[:max(0, len(buf) - 2)]
                    break
                if s:
                    del buf[:s]
            try:
                e = jpeg_end(buf, 0, state)
            except ValueError:
                del buf[:2]                        # not a frame after all: look for the next SOI
                state.clear()
                continue
            if e is None:
                if len(buf) > max_frame_bytes:     # no end in sight: drop it, resynchronise
                    del buf[:2]
                    state.clear()
                    continue
                break
            state.clear()
            yield JpegFrame(bytes(buf[:e]))
            del buf[:e]


def _chunks(fp: BinaryIO, size: int = 1 << 16) -> Iterator[bytes]:
    while True:
        b = fp.read(size)
        if not b:
            return
        yield b


def multipart_frames(fp: BinaryIO, max_frame_bytes: int = MAX_FRAME_BYTES) -> Iterator[JpegFrame]:
    """body of an HTTP `multipart/x-mixed-replace` response (IP cameras): parts with a Content-Length are cut by it, parts
    without one by the JPEG's own end-of-image; anything that is not a JPEG part is skipped.  A Content-Length that is negative
    or larger than `max_frame_bytes` is not believed (the part is then cut by its markers), and a part without an end is dropped
    once it has grown past that size"""
    buf = bytearray()
    eof = False

    def need(n: int) -> bool:
        nonlocal eof
        while len(buf) < n and not eof:
            b = fp.read(max(1 << 16, n - len(buf)))
            if not b:
                eof = True
            else:
                buf.extend(b)
        return len(buf) >= n

    while True:
        # headers of the next part end with an empty line
        while True:
            h = buf.find(b"\r\n\r\n")
            s = buf.find(b"\xff\xd8\xff")
            if h >= 0 and (s < 0 or h < s):
                break
            if s >= 0:
                h = -1
                break
            if not need(len(buf) + 1):
                return
        length = None
        if h >= 0:
            for line in bytes(buf[:h]).split(b"\r\n"):
                k, _, v = line.partition(b":")
                if k.strip().lower() == b"content-length":
                    try:
                        length = int(v.strip())
                    except ValueError:
                        length = None
                    if length is not None and not (0 < length <= max_frame_bytes):
                        length = None
            del buf[:h + 4]
        if length is not None:
            if not need(length):
                return
            part = bytes(buf[:length])
            del buf[:length]
            if part[:3] == b"\xff\xd8\xff":
                yield JpegFrame(part)
            continue
        # no length: the frame ends at its own EOI
        state: list = []
        while True:
            if not state:
                s = buf.find(b"\xff\xd8\xff")
                if s < 0:
                    del buf[:max(0, len(buf) - 2)]
                    if not need(len(buf) + 1):
                        return
                    continue
                del buf[:s]
            try:
                e = jpeg_end(buf, 0, state)
            except ValueError:
                del buf[:2]
                state.clear()
                continue
            if e is not None:
                yield JpegFrame(bytes(buf[:e]))
                del buf[:e]
                break
            if len(buf) > max_frame_bytes:         # no end in sight: drop it, resynchronise
                del buf[:2]
                state.clear()
                continue
            if not need(len(buf) + 1):
                return


def _skip(fp: Any, n: int) -> None:
    if n <= 0:
        return
    try:
        fp.seek(n, 1)
    except (AttributeError, OSError, io.UnsupportedOperation):
        while n > 0:
            got = fp.read(min(n, 1 << 16))
            if not got:
                return
            n -= len(got)


def avi_frames(fp: BinaryIO, max_frame_bytes: int = MAX_FRAME_BYTES) -> Iterator[JpegFrame]:
    """video chunks ('..dc' / '..db') of the first MJPG stream of a RIFF/AVI file, in file order (the lists that hold chunks -
    hdrl, strl, movi, rec - are walked through, everything else is skipped by its length; no index is needed).  Raises
    ValueError for a file that is not an AVI or whose video stream is not Motion-JPEG.  Chunk sizes come from the file: a video
    chunk larger than `max_frame_bytes` is skipped, a stream header is read up to 64 KiB."""
    head = fp.read(12)
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"AVI ":
        raise ValueError("not a RIFF/AVI file")
    video_stream = None
    n_streams = 0
    while True:
        hdr = fp.read(8)
        if len(hdr) < 8:
            return
        cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
        if cid == b"LIST":
            kind = fp.read(4)
            if kind not in (b"hdrl", b"strl", b"movi", b"rec "):
                _skip(fp, size - 4 + (size & 1))
        elif cid == b"RIFF":                       # OpenDML continuation ('AVIX'): more movi lists
            fp.read(4)
        elif cid == b"strh":
            data = fp.read(min(size, 1 << 16))
            _skip(fp, size + (size & 1) - len(data))
            if data[:4] == b"vids" and video_stream is None:
                fourcc = data[4:8].upper()
                if fourcc not in (b"MJPG", b"JPEG", b"AVRN", b"LJPG", b"\0\0\0\0"):
                    raise ValueError(f"AVI video stream is {fourcc!r}, not Motion-JPEG")
                video_stream = n_streams
            n_streams += 1
        elif cid[2:] in (b"dc", b"db") and cid[:2].isdigit():
            if size > max_frame_bytes:
                _skip(fp, size + (size & 1))
                continue
            data = fp.read(size + (size & 1))
            if (video_stream is None or int(cid[:2]) == video_stream) and data[:3] == b"\xff\xd8\xff":
                yield JpegFrame(data[:size])
        else:
            _skip(fp, size + (size & 1))


class _Prefixed:
    """a binary source whose first bytes were read for sniffing: those bytes again, then the rest"""

    def __init__(self, head: bytes, rest: Any):
        self._h, self._rest = head, rest

    def read(self, n: int = -1) -> bytes:
        if n is None or n < 0:
            out, self._h = self._h + self._rest.read(), b""
            return out
        out, self._h = self._h[:n], self._h[n:]
        if len(out) < n:
            out += self._rest.read(n - len(out))
        return out

    def seek(self, off: int, whence: int = 0):
        if whence != 1 or off < 0:
            raise io.UnsupportedOperation("only forward relative seeks")
        k = min(off, len(self._h))
        self._h = self._h[k:]
        if off > k:
            self._rest.seek(off - k, 1)


def open_frames(source: Any, container: str = "auto") -> Iterator[JpegFrame]:
    """frames of `source`: a path, a binary file object, or an iterable of byte chunks; container "raw" (back-to-back JPEGs),
    "multipart", "avi" or "auto" (by the first bytes)"""
    fp: Any = open(source, "rb") if isinstance(source, str) else source
    if not hasattr(fp, "read"):
        if container not in ("auto", "raw"):
            raise ValueError("an iterable of chunks can only be a raw stream")
        return split_stream(fp)
    if container == "auto":
        head = fp.read(16)
        container = "avi" if head[:4] == b"RIFF" else ("raw" if head[:2] == b"\xff\xd8" else "multipart")
        fp = _Prefixed(head, fp)
    if container == "avi":
        return avi_frames(fp)
    if container == "multipart":
        return multipart_frames(fp)
    if container == "raw":
        return split_stream(_chunks(fp))
    raise ValueError(f"unknown container {container!r}")


class MjpegCapture:
    """cv2.VideoCapture's surface (camera.py:185-221: isOpened / read / open / release) over a Motion-JPEG source; `read()`
    returns (True, JpegFrame) - the frame stays encoded - or (False, None) at the end of the stream.  `opener` is called for
    every (re)open and returns a path, a binary file object or an iterable of chunks (an HTTP response body)."""

    def __init__(self, opener, container: str = "auto"):
        self._opener = opener if callable(opener) else (lambda: opener)
        self._container = container
        self._it: Optional[Iterator[JpegFrame]] = None
        self.frames_read = 0
        self.open()

    def open(self, *_a) -> bool:
        try:
            self._it = iter(open_frames(self._opener(), self._container))
        except (OSError, ValueError):
            self._it = None
        return self._it is not None

    def isOpened(self) -> bool:
        return self._it is not None

    def read(self) -> Tuple[bool, Optional[JpegFrame]]:
        if self._it is None:
            return False, None
        try:
            f = next(self._it)
        except StopIteration:
            return False, None
        except (OSError, ValueError):
            self._it = None                        # a broken source counts as a closed capture (one reopen attempt, camera.py:185-200)
            return False, None
        self.frames_read += 1
        return True, f

    def release(self) -> None:
        self._it = None


# ---------------------------------------------------------------------------------------------- writers (tests, tools)

def write_avi(frames: Sequence[bytes], hw: Tuple[int, int], fps: int = 25) -> bytes:
    """a minimal RIFF/AVI file with one MJPG video stream (what a USB-camera recorder writes): tests and tools only"""
    def chunk(cid: bytes, data: bytes) -> bytes:
        return cid + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(kind: bytes, body: bytes) -> bytes:
        return b"LIST" + struct.pack("<I", len(body) + 4) + kind + body

    h, w = hw
    avih = struct.pack("<14I", 1000000 // fps, 0, 0, 0x10, len(frames), 0, 1, max(map(len, frames), default=0), w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, len(frames), 0, 0xFFFFFFFF, 0) + struct.pack("<4H", 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi = lst(b"movi", b"".join(chunk(b"00dc", bytes(f)) for f in frames))
    body = b"AVI " + hdrl + chunk(b"JUNK", b"\0" * 12) + movi
    return b"RIFF" + struct.pack("<I", len(body)) + body


def write_multipart(frames: Sequence[bytes], boundary: bytes = b"frpframe", with_length: bool = True) -> bytes:
    """the body an IP camera sends for `multipart/x-mixed-replace; boundary=...`: tests and tools only"""
    out = bytearray()
    for f in frames:
        out += b"--" + boundary + b"\r\nContent-Type: image/jpeg\r\n"
        if with_length:
            out += b"Content-Length: " + str(len(f)).encode() + b"\r\n"
        out += b"\r\n" + bytes(f) + b"\r\n"
    return bytes(out)
