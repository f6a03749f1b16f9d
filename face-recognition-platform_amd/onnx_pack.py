"""Real-weight loader for ONNX model packs, without the `onnx` / `protobuf` packages.

The reference gets its networks from downloaded model packs (insightface `FaceAnalysis`,
backend/app/utils/deepfake_utils.py:39-51; dlib .dat files behind face_recognition,
backend/app/services/face_service.py:156,179).  This module reads a user-supplied `.onnx` file
(protobuf wire format parsed by hand), and maps an ArcFace IResNet graph and a detector graph onto the raw
dict that `weights.pack_blob` consumes:

* `parse_model(bytes)`      -> Graph(nodes, initializers)            (generic, any ONNX file)
* `raw_from_onnx(path)`     -> {"emb.conv1.weight": ..., ...}        (IResNet-50/100 family)
* `det_raw_from_onnx(path)` -> {"det.stem1.conv.weight": ..., ...}   (the FRPDet family, below)
* `pack_from_onnx(det, emb)`-> weight blob bytes for frp_load_weights
* `write_model(...)`        -> bytes                                  (minimal writer; `iresnet_to_onnx` /
  `detector_to_onnx` use it to produce files in the layouts exporters emit: tests and interchange)

THE DETECTOR.  The only detector this repo runs is its own FRPDet (netspec.detector_layers: two stride-2 stems, four
stages of basic residual blocks, a three-level top-down FPN with nearest 2x upsampling, 3x3 smoothing, two shared-shape
tower convs and a 3x3 output conv per level, 2 anchors x (score logit, 4 distances, 10 landmark offsets) - the SCRFD /
RetinaFace head convention).  A public SCRFD pack is NOT loadable: its backbone / PAFPN topology differs and no such
file exists offline to map it against; what ships is the interchange format for FRPDet weights trained elsewhere:
`detector_to_onnx` writes the graph, `det_raw_from_onnx` reads it back by parameter name or - for anonymous
initializers - by walking the DATAFLOW (which tensor feeds which node), so node order, BatchNorm folding, epsilon
convention, shortcut-first emission, laterals-first emission and SCRFD-style split heads (separate score / bbox /
landmark convs per level, optionally followed by a Sigmoid on the scores) are all understood.

Two layouts are understood:
  1. named: initializers carry this repo's / arcface_torch's parameter names
     (`conv1.weight`, `layer1.0.bn1.running_mean`, ... with or without a prefix) -> taken by name;
  2. anonymous: numeric tensor names as written by `torch.onnx.export` + simplifiers.  The node list
     (topologically ordered by the ONNX spec) is walked as a stream of Conv / BatchNormalization /
     PRelu / Add / Gemm|MatMul events and matched against the IResNet block structure
     (BN -> conv3x3 -> [BN] -> PReLU -> conv3x3(stride) -> [BN] -> [conv1x1(stride) -> [BN]] -> Add);
     a BatchNorm the exporter folded into the preceding Conv/Gemm (bias present, BN absent) becomes an
     identity BN whose beta is that bias, so `weights.fold_layer` and the oracle need no special case.
BatchNormalization `epsilon` other than weights.BN_EPS is absorbed into the stored variance.
No model file exists offline: the layouts are exercised on files written by `write_model`.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import netspec as ns
from .weights import BN_EPS

# ----------------------------------------------------------------------------- protobuf wire format


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _fields(buf: bytes):
    """yield (field number, wire type, value) with value = int (wire 0/1/5 raw) or bytes (wire 2)"""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > n:
                raise ValueError("truncated fixed64")
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            if pos + 4 > n:
                raise ValueError("truncated fixed32")
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v, wt) -> List[int]:
    if wt == 0:
        return [_signed64(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64,
           9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    """TensorProto -> (name, array)"""
    dims: List[int] = []
    dtype = 1
    name = ""
    raw: Optional[bytes] = None
    floats: List[float] = []
    ints: List[int] = []
    doubles: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 4:                                   # float_data (packed or repeated fixed32)
            floats += list(np.frombuffer(v, "<f4"))
        elif fno in (5, 7, 11):                          # int32_data / int64_data / uint64_data
            ints += _packed_varints(v, wt)
        elif fno == 8:
            name = v.decode("utf-8", "replace")
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            doubles += list(np.frombuffer(v, "<f8"))
        elif fno == 13 or fno == 14:
            raise ValueError(f"tensor {name!r}: external data is not supported (save the model with embedded weights)")
    if dtype not in _DTYPES:
        raise ValueError(f"tensor {name!r}: unsupported data type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dt.newbyteorder("<")).astype(dt)
    elif floats:
        arr = np.asarray(floats, np.float32).astype(dt)
    elif doubles:
        arr = np.asarray(doubles, np.float64).astype(dt)
    elif dtype == 10:                                    # fp16 travels as uint16 bit patterns in int32_data
        arr = np.asarray(ints, np.int64).astype(np.uint16).view(np.float16)
    else:
        arr = np.asarray(ints, np.int64).astype(dt)
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise ValueError(f"tensor {name!r}: {arr.size} elements for dims {dims}")
    return name, arr.reshape(dims) if dims else arr.reshape(())


@dataclass
class Node:
    op: str
    inputs: List[str]
    outputs: List[str]
    name: str = ""
    attrs: Dict[str, object] = field(default_factory=dict)


@dataclass
class Graph:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[str]
    outputs: List[str]


def _attribute(buf: bytes) -> Tuple[str, object]:
    name, val = "", None
    ints: List[int] = []
    floats: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(np.frombuffer(v, "<f4"))
        elif fno == 8:
            ints += _packed_varints(v, wt)
    if val is None:
        val = ints if ints else (floats if floats else None)
    return name, val


def _node(buf: bytes) -> Node:
    n = Node("", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(v.decode())
        elif fno == 2:
            n.outputs.append(v.decode())
        elif fno == 3:
            n.name = v.decode()
        elif fno == 4:
            n.op = v.decode()
        elif fno == 5:
            k, a = _attribute(v)
            n.attrs[k] = a
    return n


def _value_info_name(buf: bytes) -> str:
    for fno, wt, v in _fields(buf):
        if fno == 1:
            return v.decode()
    return ""


def parse_model(buf: bytes) -> Graph:
    """ModelProto bytes -> Graph.  `Constant` nodes are lifted into the initializer table."""
    graph = None
    for fno, wt, v in _fields(buf):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError("not an ONNX ModelProto (no graph field)")
    g = Graph([], {}, [], [])
    for fno, wt, v in _fields(graph):
        if fno == 1:
            g.nodes.append(_node(v))
        elif fno == 5:
            name, arr = _tensor(v)
            g.initializers[name] = arr
        elif fno == 11:
            g.inputs.append(_value_info_name(v))
        elif fno == 12:
            g.outputs.append(_value_info_name(v))
    for n in g.nodes:
        if n.op == "Constant" and n.outputs and isinstance(n.attrs.get("value"), np.ndarray):
            g.initializers[n.outputs[0]] = n.attrs["value"]
    g.inputs = [i for i in g.inputs if i not in g.initializers]
    return g


# ----------------------------------------------------------------------------- minimal writer


def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc(fno: int, wt: int, payload) -> bytes:
    tag = _enc_varint((fno << 3) | wt)
    if wt == 0:
        return tag + _enc_varint(payload)
    if wt == 2:
        return tag + _enc_varint(len(payload)) + bytes(payload)
    return tag + bytes(payload)


_DT_CODE = {np.dtype(v): k for k, v in _DTYPES.items()}


def _enc_tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    arr = np.asarray(arr)
    out = b"".join(_enc(1, 0, int(d)) for d in arr.shape)                  # unpacked dims (both forms are legal)
    out += _enc(2, 0, _DT_CODE[arr.dtype])
    if raw:
        out += _enc(9, 2, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    elif arr.dtype == np.float32:
        out += _enc(4, 2, arr.astype("<f4").tobytes())                     # packed float_data
    elif arr.dtype == np.float64:
        out += _enc(10, 2, arr.astype("<f8").tobytes())                    # packed double_data
    elif arr.dtype == np.float16:                                          # bit patterns in int32_data
        out += _enc(5, 2, b"".join(_enc_varint(int(x)) for x in arr.reshape(-1).view(np.uint16)))
    else:
        out += _enc(7 if arr.dtype == np.int64 else 5, 2, b"".join(_enc_varint(int(x)) for x in arr.reshape(-1)))
    out += _enc(8, 2, name.encode())
    return out


def _enc_attr(name: str, val) -> bytes:
    out = _enc(1, 2, name.encode())
    if isinstance(val, float):
        out += _enc(2, 5, struct.pack("<f", val)) + _enc(20, 0, 1)
    elif isinstance(val, int):
        out += _enc(3, 0, val) + _enc(20, 0, 2)
    elif isinstance(val, (bytes, str)):
        out += _enc(4, 2, val.encode() if isinstance(val, str) else val) + _enc(20, 0, 3)
    elif isinstance(val, np.ndarray):
        out += _enc(5, 2, _enc_tensor("", val)) + _enc(20, 0, 4)
    else:
        out += _enc(8, 2, b"".join(_enc_varint(int(x)) for x in val)) + _enc(20, 0, 7)   # packed ints
    return out


def write_model(nodes: List[Node], initializers: Dict[str, np.ndarray], inputs: List[str], outputs: List[str],
                raw_data: bool = True) -> bytes:
    g = b""
    for n in nodes:
        nb = b"".join(_enc(1, 2, i.encode()) for i in n.inputs) + b"".join(_enc(2, 2, o.encode()) for o in n.outputs)
        nb += _enc(3, 2, n.name.encode()) + _enc(4, 2, n.op.encode())
        nb += b"".join(_enc(5, 2, _enc_attr(k, v)) for k, v in n.attrs.items())
        g += _enc(1, 2, nb)
    g += _enc(2, 2, b"frp")
    for name, arr in initializers.items():
        g += _enc(5, 2, _enc_tensor(name, arr, raw_data))
    for i in inputs:
        g += _enc(11, 2, _enc(1, 2, i.encode()))
    for o in outputs:
        g += _enc(12, 2, _enc(1, 2, o.encode()))
    model = _enc(1, 0, 8) + _enc(2, 2, b"frp_amd.onnx_pack") + _enc(7, 2, g)
    model += _enc(8, 2, _enc(1, 2, b"") + _enc(2, 0, 13))                  # opset_import {domain "", version 13}
    return model


# ----------------------------------------------------------------------------- IResNet mapping

_BN_FIELDS = ("weight", "bias", "running_mean", "running_var")


def _put_bn(raw: Dict[str, np.ndarray], name: str, g, b, m, v, eps: float):
    raw[name + ".weight"] = np.asarray(g, np.float32).reshape(-1)
    raw[name + ".bias"] = np.asarray(b, np.float32).reshape(-1)
    raw[name + ".running_mean"] = np.asarray(m, np.float32).reshape(-1)
    # fold_layer / the oracle use weights.BN_EPS: keep var + eps what the file says
    raw[name + ".running_var"] = (np.asarray(v, np.float64).reshape(-1) + (eps - BN_EPS)).astype(np.float32)


def _identity_bn(raw: Dict[str, np.ndarray], name: str, c: int, beta=None):
    raw[name + ".weight"] = np.ones(c, np.float32)
    raw[name + ".bias"] = np.zeros(c, np.float32) if beta is None else np.asarray(beta, np.float32).reshape(-1)
    raw[name + ".running_mean"] = np.zeros(c, np.float32)
    raw[name + ".running_var"] = np.full(c, 1.0 - BN_EPS, np.float64).astype(np.float32)


def _by_name(init: Dict[str, np.ndarray]) -> Optional[Dict[str, np.ndarray]]:
    """layout 1: parameter names survive in the file"""
    key = next((k for k in init if k.endswith("layer1.0.conv1.weight")), None)
    if key is None:
        return None
    prefix = key[:-len("layer1.0.conv1.weight")]
    raw = {}
    for k, v in init.items():
        if not k.startswith(prefix) or k.endswith("num_batches_tracked"):
            continue
        kk = k[len(prefix):]
        v = np.asarray(v, np.float32)
        if kk.endswith("prelu.weight"):
            v = v.reshape(-1)
        raw["emb." + kk] = v
    return raw


def _events(g: Graph):
    """Conv / BN / PRelu / Add / FC events in node order; everything else is shape plumbing"""
    init = g.initializers
    ev = []
    for n in g.nodes:
        if n.op == "Conv":
            w = init[n.inputs[1]]
            b = init[n.inputs[2]] if len(n.inputs) > 2 and n.inputs[2] else None
            strides = n.attrs.get("strides") or [1, 1]
            ev.append(("conv", np.asarray(w, np.float32), None if b is None else np.asarray(b, np.float32), int(strides[0])))
        elif n.op == "BatchNormalization":
            eps = float(n.attrs.get("epsilon", 1e-5))
            ev.append(("bn", [np.asarray(init[i], np.float32) for i in n.inputs[1:5]], eps))
        elif n.op == "PRelu":
            ev.append(("prelu", np.asarray(init[n.inputs[1]], np.float32).reshape(-1)))
        elif n.op == "Add":
            if n.inputs[0] in init or n.inputs[1] in init:          # MatMul + Add(bias)
                ev.append(("bias", np.asarray(init[n.inputs[0]] if n.inputs[0] in init else init[n.inputs[1]], np.float32)))
            else:
                ev.append(("add",))
        elif n.op == "Gemm":
            w = np.asarray(init[n.inputs[1]], np.float32)
            if not int(n.attrs.get("transB", 0)):
                w = w.T
            alpha, beta = float(n.attrs.get("alpha", 1.0)), float(n.attrs.get("beta", 1.0))
            b = np.asarray(init[n.inputs[2]], np.float32) * beta if len(n.inputs) > 2 else np.zeros(w.shape[0], np.float32)
            ev.append(("fc", w * alpha, b))
        elif n.op == "MatMul" and n.inputs[1] in init:
            ev.append(("fc", np.asarray(init[n.inputs[1]], np.float32).T, None))
    return ev


def _structural(g: Graph) -> Dict[str, np.ndarray]:
    """layout 2: walk the event stream against the IResNet structure"""
    ev = _events(g)
    pos = 0

    def peek(kind):
        return pos < len(ev) and ev[pos][0] == kind

    def take(kind, what):
        nonlocal pos
        if not peek(kind):
            got = ev[pos][0] if pos < len(ev) else "end of graph"
            raise ValueError(f"ONNX graph does not look like an ArcFace IResNet: expected {kind} for {what}, found {got}")
        pos += 1
        return ev[pos - 1]

    raw: Dict[str, np.ndarray] = {}

    def conv_bn(dst: Dict[str, np.ndarray], name: str, bn_name: str, ksize: int, before_block: bool = False):
        _, w, b, stride = take("conv", name)
        if w.shape[2] != ksize:
            raise ValueError(f"{name}: expected a {ksize}x{ksize} kernel, file has {w.shape[2]}x{w.shape[3]}")
        dst[name + ".weight"] = w
        # a leading shortcut is followed by the block's own bn1: the BN after it is the shortcut's only
        # if a second BN follows
        own_bn = peek("bn") and (not before_block or (pos + 1 < len(ev) and ev[pos + 1][0] == "bn"))
        if own_bn:
            _, p, eps = take("bn", bn_name)
            if b is not None:                       # conv bias in front of a live BN: push it through the mean
                p = [p[0], p[1], p[2] - b, p[3]]
            _put_bn(dst, bn_name, *p, eps)
        else:
            _identity_bn(dst, bn_name, w.shape[0], b)

    def is_1x1_conv():
        return peek("conv") and ev[pos][1].shape[2] == 1

    conv_bn(raw, "emb.conv1", "emb.bn1", 3)
    raw["emb.prelu.weight"] = take("prelu", "emb.prelu")[1]
    li, bi = 1, 0
    while True:
        lead_ds = is_1x1_conv()                     # some exporters emit the shortcut branch first
        if not lead_ds and not (peek("bn") and pos + 1 < len(ev) and ev[pos + 1][0] == "conv"):
            break
        blk: Dict[str, np.ndarray] = {}
        if lead_ds:
            conv_bn(blk, "downsample.0", "downsample.1", 1, before_block=True)
        _, p, eps = take("bn", "block bn1")
        _put_bn(blk, "bn1", *p, eps)
        conv_bn(blk, "conv1", "bn2", 3)
        blk["prelu.weight"] = take("prelu", "block prelu")[1]
        conv_bn(blk, "conv2", "bn3", 3)
        if not lead_ds and is_1x1_conv():
            conv_bn(blk, "downsample.0", "downsample.1", 1)
        take("add", "residual add")
        if "downsample.0.weight" in blk and bi != 0:  # a block with a shortcut conv opens the next stage
            li, bi = li + 1, 0
        for k, v in blk.items():
            raw[f"emb.layer{li}.{bi}.{k}"] = v
        bi += 1
    _, p, eps = take("bn", "emb.bn2")
    _put_bn(raw, "emb.bn2", *p, eps)
    _, w, b = take("fc", "emb.fc")
    if b is None and peek("bias"):
        b = take("bias", "emb.fc.bias")[1]
    raw["emb.fc.weight"] = w
    fc_b = np.zeros(w.shape[0], np.float32) if b is None else b.reshape(-1)
    if peek("bn"):
        _, p, eps = take("bn", "emb.features")
        raw["emb.fc.bias"] = fc_b
        _put_bn(raw, "emb.features", *p, eps)
    else:                                           # features BN folded into the Gemm
        raw["emb.fc.bias"] = fc_b
        _identity_bn(raw, "emb.features", w.shape[0])
    return raw


def raw_from_onnx(path_or_bytes) -> Dict[str, np.ndarray]:
    """ArcFace IResNet `.onnx` -> raw dict in `weights.make_synthetic_raw` naming (emb.* keys only).
    Raises ValueError with the first structural mismatch; validates every tensor shape against
    `netspec.iresnet_layers` of the block counts found."""
    buf = path_or_bytes
    if not isinstance(buf, (bytes, bytearray, memoryview)):
        with open(path_or_bytes, "rb") as f:
            buf = f.read()
    g = parse_model(bytes(buf))
    raw = _by_name(g.initializers) or _structural(g)
    from .weights import emb_blocks_of
    blocks = emb_blocks_of(raw)
    if min(blocks) < 1:
        raise ValueError(f"ONNX graph does not look like an ArcFace IResNet: stages {blocks}")
    for l in ns.iresnet_layers(blocks):
        cin, cout = l.cin_real or l.cin, l.cout_real or l.cout
        w = raw.get(l.name + ".weight")
        want = (cout, cin) if l.name == "emb.fc" else (cout, cin, l.k, l.k)
        if w is None or tuple(w.shape) != want:
            raise ValueError(f"{l.name}.weight: expected shape {want}, file has {None if w is None else tuple(w.shape)}")
        for bn, c in ((l.pre_bn, cin if l.name != "emb.fc" else 512), (l.post_bn, cout)):
            if bn:
                for f_ in _BN_FIELDS:
                    if raw.get(f"{bn}.{f_}") is None or raw[f"{bn}.{f_}"].shape != (c,):
                        raise ValueError(f"{bn}.{f_}: missing or not of shape ({c},)")
        if l.prelu and raw.get(l.prelu + ".weight", np.zeros(0)).shape != (cout,):
            raise ValueError(f"{l.prelu}.weight: missing or not of shape ({cout},)")
    return raw


# ----------------------------------------------------------------------------- exporter (tests, interchange)


def iresnet_to_onnx(raw: Dict[str, np.ndarray], named: bool = True, fuse_bn: bool = False, eps: float = BN_EPS,
                    raw_data: bool = True, shortcut_first: bool = False, matmul_fc: bool = False) -> bytes:
    """Write the embedder of a raw dict as an ONNX file.  `named=False` uses numeric tensor names,
    `fuse_bn=True` folds every BatchNorm that FOLLOWS a Conv/Gemm into it (what eval-mode exporters and
    simplifiers do; the pre-conv bn1 of each block stays a node), `eps` is written as the BatchNormalization
    epsilon with the variance shifted so the function is unchanged, `shortcut_first` emits the 1x1 shortcut
    before the main branch, `matmul_fc` writes the FC as MatMul + Add instead of Gemm."""
    from .weights import emb_blocks_of
    blocks = emb_blocks_of(raw)
    nodes: List[Node] = []
    init: Dict[str, np.ndarray] = {}
    counter = [0]

    def tname(n: str) -> str:
        if named:
            return n[len("emb."):]
        counter[0] += 1
        return str(1000 + counter[0])

    def act() -> str:
        counter[0] += 1
        return f"t{counter[0]}"

    def bn_params(name: str):
        g, b, m, v = (raw[f"{name}.{f_}"].astype(np.float64) for f_ in _BN_FIELDS)
        return g, b, m, v

    def emit_bn(x: str, name: str) -> str:
        g, b, m, v = bn_params(name)
        names = []
        for f_, arr in zip(_BN_FIELDS, (g, b, m, v + (BN_EPS - eps))):
            tn = tname(f"{name}.{f_}")
            init[tn] = arr.astype(np.float32)
            names.append(tn)
        y = act()
        nodes.append(Node("BatchNormalization", [x] + names, [y], name, {"epsilon": float(eps)}))
        return y

    def emit_conv(x: str, name: str, bn: str, stride: int, k: int) -> str:
        w = raw[name + ".weight"].astype(np.float64)
        ins = [x]
        if fuse_bn:
            g, b, m, v = bn_params(bn)
            s = g / np.sqrt(v + BN_EPS)
            wn, bname = tname(name + ".weight"), tname(name + ".bias")
            init[wn] = (w * s[:, None, None, None]).astype(np.float32)
            init[bname] = (b - m * s).astype(np.float32)
            ins += [wn, bname]
        else:
            wn = tname(name + ".weight")
            init[wn] = w.astype(np.float32)
            ins.append(wn)
        y = act()
        nodes.append(Node("Conv", ins, [y], name, {"kernel_shape": [k, k], "strides": [stride, stride],
                                                    "pads": [k // 2] * 4, "dilations": [1, 1], "group": 1}))
        return y if fuse_bn else emit_bn(y, bn)

    def emit_prelu(x: str, name: str) -> str:
        tn = tname(name + ".weight")
        init[tn] = raw[name + ".weight"].astype(np.float32).reshape(-1, 1, 1)
        y = act()
        nodes.append(Node("PRelu", [x, tn], [y], name))
        return y

    x = emit_prelu(emit_conv("data", "emb.conv1", "emb.bn1", 1, 3), "emb.prelu")
    for li, nb in enumerate(blocks, start=1):
        for bi in range(nb):
            p = f"emb.layer{li}.{bi}"
            stride = 2 if bi == 0 else 1
            ident = x
            if bi == 0 and shortcut_first:
                ident = emit_conv(x, p + ".downsample.0", p + ".downsample.1", stride, 1)
            y = emit_bn(x, p + ".bn1")
            y = emit_prelu(emit_conv(y, p + ".conv1", p + ".bn2", 1, 3), p + ".prelu")
            y = emit_conv(y, p + ".conv2", p + ".bn3", stride, 3)
            if bi == 0 and not shortcut_first:
                ident = emit_conv(x, p + ".downsample.0", p + ".downsample.1", stride, 1)
            x = act()
            nodes.append(Node("Add", [y, ident], [x], p + ".add"))
    x = emit_bn(x, "emb.bn2")
    y = act()
    nodes.append(Node("Flatten", [x], [y], "flatten", {"axis": 1}))
    w, b = raw["emb.fc.weight"].astype(np.float64), raw["emb.fc.bias"].astype(np.float64)
    if fuse_bn:
        g, bb, m, v = bn_params("emb.features")
        s = g / np.sqrt(v + BN_EPS)
        w, b = w * s[:, None], (b - m) * s + bb
    wn, bname = tname("emb.fc.weight"), tname("emb.fc.bias")
    z = act()
    if matmul_fc:
        init[wn], init[bname] = w.T.astype(np.float32), b.astype(np.float32)
        nodes.append(Node("MatMul", [y, wn], [z], "fc"))
        z2 = act()
        nodes.append(Node("Add", [z, bname], [z2], "fc.bias"))
        z = z2
    else:
        init[wn], init[bname] = w.astype(np.float32), b.astype(np.float32)
        nodes.append(Node("Gemm", [y, wn, bname], [z], "fc", {"alpha": 1.0, "beta": 1.0, "transB": 1}))
    out = z if fuse_bn else emit_bn(z, "emb.features")
    return write_model(nodes, init, ["data"], [out], raw_data)


# ----------------------------------------------------------------------------- detector (FRPDet) mapping


def _attr_str(n: Node, key: str, default: str = "") -> str:
    v = n.attrs.get(key, default)
    return v.decode() if isinstance(v, (bytes, bytearray)) else str(v)


class _Flow:
    """producer / consumer lookup over a parsed graph (initializers are not tensors of the dataflow)"""

    def __init__(self, g: Graph):
        self.g = g
        self.cons: Dict[str, List[Node]] = {}
        for n in g.nodes:
            for i in n.inputs:
                if i and i not in g.initializers:
                    self.cons.setdefault(i, []).append(n)

    def consumers(self, t: str, op: Optional[str] = None) -> List[Node]:
        return [n for n in self.cons.get(t, []) if op is None or n.op == op]

    def conv_params(self, n: Node):
        init = self.g.initializers
        w = np.asarray(init[n.inputs[1]], np.float32)
        b = np.asarray(init[n.inputs[2]], np.float32) if len(n.inputs) > 2 and n.inputs[2] else None
        stride = int((n.attrs.get("strides") or [1, 1])[0])
        if int(n.attrs.get("group", 1)) != 1:
            raise ValueError(f"{n.name or n.outputs[0]}: grouped convolutions are not part of the FRPDet family")
        return w, b, stride

    def unit(self, conv: Node, raw: Dict[str, np.ndarray], name: str, bn_name: Optional[str], want_relu: bool, k: int, what: str):
        """Conv [-> BatchNormalization] [-> Relu] starting at `conv`: stores the conv (and its BN, or an identity BN carrying the
        folded bias) under `name` / `bn_name`, returns (output tensor, stride).  bn_name None: a conv with its own bias, no BN."""
        w, b, stride = self.conv_params(conv)
        if w.shape[2] != k or w.shape[3] != k:
            raise ValueError(f"{what}: expected a {k}x{k} kernel, file has {w.shape[2]}x{w.shape[3]}")
        raw[name + ".weight"] = w
        t = conv.outputs[0]
        bn = self.consumers(t, "BatchNormalization")
        if bn_name is None:
            if bn:
                raise ValueError(f"{what}: unexpected BatchNormalization behind a biased conv")
            raw[name + ".bias"] = np.zeros(w.shape[0], np.float32) if b is None else b.reshape(-1)
        elif bn:
            p = [np.asarray(self.g.initializers[i], np.float32) for i in bn[0].inputs[1:5]]
            if b is not None:
                p = [p[0], p[1], p[2] - b, p[3]]
            _put_bn(raw, bn_name, *p, float(bn[0].attrs.get("epsilon", 1e-5)))
            t = bn[0].outputs[0]
        else:
            _identity_bn(raw, bn_name, w.shape[0], b)
        if want_relu:
            r = self.consumers(t, "Relu")
            if not r:
                raise ValueError(f"{what}: expected a Relu behind the conv, found {[n.op for n in self.consumers(t)] or 'the end of the graph'}")
            t = r[0].outputs[0]
        return t, stride


def _is_k(flow: _Flow, n: Node, k: int) -> bool:
    return n.op == "Conv" and flow.g.initializers[n.inputs[1]].shape[2] == k


def _det_by_name(init: Dict[str, np.ndarray]) -> Optional[Dict[str, np.ndarray]]:
    key = next((k for k in init if k.endswith("stem1.conv.weight")), None)
    if key is None:
        return None
    prefix = key[:-len("stem1.conv.weight")]
    raw = {}
    for k, v in init.items():
        if k.startswith(prefix) and k.rsplit(".", 1)[-1] in ("weight", "bias", "running_mean", "running_var"):
            raw["det." + k[len(prefix):]] = np.asarray(v, np.float32)
    return raw


def _det_structural(g: Graph) -> Dict[str, np.ndarray]:
    if len(g.inputs) != 1:
        raise ValueError(f"detector graph must have one input, file has {g.inputs}")
    fl = _Flow(g)
    raw: Dict[str, np.ndarray] = {}

    def only_conv(t: str, k: int, what: str) -> Node:
        c = [n for n in fl.consumers(t, "Conv") if _is_k(fl, n, k)]
        if len(c) != 1:
            raise ValueError(f"ONNX graph does not look like an FRPDet detector: expected one {k}x{k} conv for {what}, found {len(c)}")
        return c[0]

    x, s = fl.unit(only_conv(g.inputs[0], 3, "det.stem1"), raw, "det.stem1.conv", "det.stem1.bn", True, 3, "det.stem1")
    if s != 2:
        raise ValueError("det.stem1: expected stride 2")
    x, s = fl.unit(only_conv(x, 3, "det.stem2"), raw, "det.stem2.conv", "det.stem2.bn", True, 3, "det.stem2")
    if s != 2:
        raise ValueError("det.stem2: expected stride 2")
    # residual stages: a block starts at a tensor consumed by a 3x3 conv AND (an Add = identity shortcut, or a 1x1 conv of the
    # same stride whose normalised output meets the main branch in an Add)
    li, bi, feats = 1, 0, {}
    while True:
        c3 = [n for n in fl.consumers(x, "Conv") if _is_k(fl, n, 3)]
        if len(c3) != 1:
            break
        blk: Dict[str, np.ndarray] = {}
        t, stride = fl.unit(c3[0], blk, "conv1", "bn1", True, 3, "block conv1")
        c2 = [n for n in fl.consumers(t, "Conv") if _is_k(fl, n, 3)]
        if len(c2) != 1:
            break                                            # a 3x3 conv that is not a block's first: the FPN / heads begin (cannot happen
        t2, s2 = fl.unit(c2[0], blk, "conv2", "bn2", False, 3, "block conv2")       # in this family: the trunk ends in 1x1 laterals)
        add = fl.consumers(t2, "Add")
        if s2 != 1 or len(add) != 1:
            raise ValueError("ONNX graph does not look like an FRPDet detector: a block's second conv must have stride 1 and end in an Add")
        other = [i for i in add[0].inputs if i != t2][0]
        if other != x:                                       # shortcut conv: x -> Conv1x1(stride) [-> BN] -> Add
            ds = [n for n in fl.consumers(x, "Conv") if _is_k(fl, n, 1) and fl.conv_params(n)[2] == stride]
            hit = None
            for n in ds:
                tmp: Dict[str, np.ndarray] = {}
                o, _ = fl.unit(n, tmp, "downsample.0", "downsample.1", False, 1, "block shortcut")
                if o == other:
                    hit = tmp
            if hit is None:
                raise ValueError("ONNX graph does not look like an FRPDet detector: a block's Add does not meet its input or a 1x1 shortcut of it")
            blk.update(hit)
        r = fl.consumers(add[0].outputs[0], "Relu")
        if not r:
            raise ValueError("ONNX graph does not look like an FRPDet detector: no Relu behind a block's Add")
        if stride == 2:
            feats[li] = x
            li, bi = li + 1, 0
        elif stride != 1:
            raise ValueError("block stride must be 1 or 2")
        if li > 4:
            raise ValueError("ONNX graph does not look like an FRPDet detector: more than four stages")
        for k_, v in blk.items():
            raw[f"det.layer{li}.{bi}.{k_}"] = v
        bi += 1
        x = r[0].outputs[0]
    feats[li] = x
    if li != 4:
        raise ValueError(f"ONNX graph does not look like an FRPDet detector: {li} stages of residual blocks, expected 4")
    # FPN laterals: the stride-1 1x1 conv (own bias) of the outputs of stages 2, 3, 4; top-down: Add(lateral, Resize x2 nearest (level above))
    lat = {}
    for lv, src in ((3, feats[2]), (4, feats[3]), (5, feats[4])):
        c1 = [n for n in fl.consumers(src, "Conv") if _is_k(fl, n, 1) and fl.conv_params(n)[2] == 1]
        if len(c1) != 1:
            raise ValueError(f"ONNX graph does not look like an FRPDet detector: expected one lateral 1x1 conv at level {lv}, found {len(c1)}")
        lat[lv], _ = fl.unit(c1[0], raw, f"det.fpn.lat{lv}.conv", None, False, 1, f"det.fpn.lat{lv}")
    pyr = {5: lat[5]}
    for lv in (4, 3):
        add = fl.consumers(lat[lv], "Add")
        up = [n for n in fl.consumers(pyr[lv + 1]) if n.op in ("Resize", "Upsample")]
        if len(add) != 1 or len(up) != 1 or up[0].outputs[0] not in add[0].inputs:
            raise ValueError(f"ONNX graph does not look like an FRPDet detector: level {lv} is not lateral + 2x upsampled level {lv + 1}")
        if _attr_str(up[0], "mode", "nearest") != "nearest":
            raise ValueError(f"level {lv + 1} upsampling must be nearest-neighbour")
        for i in up[0].inputs[1:]:
            if i in g.initializers and g.initializers[i].size == 4 and not np.allclose(np.asarray(g.initializers[i], np.float64), [1, 1, 2, 2]):
                raise ValueError(f"level {lv + 1} upsampling must be 2x in both directions")
        pyr[lv] = add[0].outputs[0]
    A, V = ns.DET_NUM_ANCHORS, ns.DET_VALUES_PER_ANCHOR
    for lv in (3, 4, 5):
        t, _ = fl.unit(only_conv(pyr[lv], 3, f"det.fpn.smooth{lv}"), raw, f"det.fpn.smooth{lv}.conv", f"det.fpn.smooth{lv}.bn", True, 3, f"det.fpn.smooth{lv}")
        h = f"det.head{lv}"
        t, _ = fl.unit(only_conv(t, 3, f"{h}.tower0"), raw, f"{h}.tower0.conv", f"{h}.tower0.bn", True, 3, f"{h}.tower0")
        t, _ = fl.unit(only_conv(t, 3, f"{h}.tower1"), raw, f"{h}.tower1.conv", f"{h}.tower1.bn", True, 3, f"{h}.tower1")
        outs = [n for n in fl.consumers(t, "Conv") if _is_k(fl, n, 3)]
        parts = {}
        for n in outs:
            tmp: Dict[str, np.ndarray] = {}
            fl.unit(n, tmp, "o", None, False, 3, f"{h}.out")
            parts[tmp["o.weight"].shape[0]] = (tmp["o.weight"], tmp["o.bias"])
        if set(parts) == {A * V}:                            # one conv: channel = anchor * 15 + value
            raw[f"{h}.out.weight"], raw[f"{h}.out.bias"] = parts[A * V]
        elif set(parts) == {A, 4 * A, 10 * A}:               # SCRFD-style split heads: score [A], bbox [A x 4], landmarks [A x 10]
            w = np.zeros((A * V,) + parts[A][0].shape[1:], np.float32)
            b = np.zeros(A * V, np.float32)
            for a in range(A):
                w[a * V], b[a * V] = parts[A][0][a], parts[A][1][a]
                w[a * V + 1:a * V + 5], b[a * V + 1:a * V + 5] = parts[4 * A][0][4 * a:4 * a + 4], parts[4 * A][1][4 * a:4 * a + 4]
                w[a * V + 5:a * V + 15], b[a * V + 5:a * V + 15] = parts[10 * A][0][10 * a:10 * a + 10], parts[10 * A][1][10 * a:10 * a + 10]
            raw[f"{h}.out.weight"], raw[f"{h}.out.bias"] = w, b
        else:
            raise ValueError(f"{h}.out: expected one conv of {A * V} channels or three of {A} / {4 * A} / {10 * A}, file has {sorted(parts)}")
    return raw


def det_blocks_of(raw: Dict[str, np.ndarray]) -> Tuple[int, int, int, int]:
    out = []
    for li in (1, 2, 3, 4):
        n = 0
        while f"det.layer{li}.{n}.conv1.weight" in raw:
            n += 1
        out.append(n)
    return tuple(out)


def det_raw_from_onnx(path_or_bytes) -> Dict[str, np.ndarray]:
    """FRPDet `.onnx` -> raw dict in `weights.make_synthetic_raw` naming (det.* keys only).  Raises ValueError with the first
    structural mismatch; validates every tensor shape against `netspec.detector_layers` of the block counts found."""
    buf = path_or_bytes
    if not isinstance(buf, (bytes, bytearray, memoryview)):
        with open(path_or_bytes, "rb") as f:
            buf = f.read()
    g = parse_model(bytes(buf))
    raw = _det_by_name(g.initializers)
    if raw is None:
        try:
            raw = _det_structural(g)
        except ValueError as e:
            msg = str(e)
            raise ValueError(msg if "FRPDet" in msg else f"ONNX graph does not look like an FRPDet detector: {msg}") from None
    blocks = det_blocks_of(raw)
    if min(blocks) < 1:
        raise ValueError(f"ONNX graph does not look like an FRPDet detector: stages {blocks}")
    for l in ns.detector_layers(blocks):
        cin, cout = l.cin_real or l.cin, l.cout_real or l.cout
        w = raw.get(l.name + ".weight")
        if w is None or tuple(w.shape) != (cout, cin, l.k, l.k):
            raise ValueError(f"{l.name}.weight: expected shape {(cout, cin, l.k, l.k)}, file has {None if w is None else tuple(w.shape)}")
        if l.conv_bias and raw.get(l.name + ".bias", np.zeros(0)).shape != (cout,):
            raise ValueError(f"{l.name}.bias: missing or not of shape ({cout},)")
        if l.post_bn:
            for f_ in _BN_FIELDS:
                if raw.get(f"{l.post_bn}.{f_}") is None or raw[f"{l.post_bn}.{f_}"].shape != (cout,):
                    raise ValueError(f"{l.post_bn}.{f_}: missing or not of shape ({cout},)")
    return raw


def pack_from_onnx(det_onnx, emb_onnx, **pack_kwargs) -> bytes:
    """weight blob (frp_load_weights) from a detector and an embedder ONNX file"""
    from . import weights
    raw = det_raw_from_onnx(det_onnx)
    raw.update(raw_from_onnx(emb_onnx))
    return weights.pack_blob(raw, det_blocks_of(raw), weights.emb_blocks_of(raw), **pack_kwargs)


def detector_to_onnx(raw: Dict[str, np.ndarray], named: bool = True, fuse_bn: bool = False, eps: float = BN_EPS, raw_data: bool = True,
                     shortcut_first: bool = False, laterals_first: bool = False, split_heads: bool = False,
                     sigmoid_scores: bool = False) -> bytes:
    """Write the detector of a raw dict as an ONNX file (input "data" [B,3,H,W] RGB (x - 127.5) / 128; outputs the per-level head
    maps [B,30,H_l,W_l], or - `split_heads` - score / bbox / landmark maps per level as SCRFD packs have them, the scores behind
    a Sigmoid when `sigmoid_scores`).  `named=False`: numeric tensor names; `fuse_bn`: every BatchNorm folded into its conv;
    `eps`: the BatchNormalization epsilon written (variance shifted so the function is unchanged); `shortcut_first` /
    `laterals_first`: other legal node orders."""
    blocks = det_blocks_of(raw)
    nodes: List[Node] = []
    init: Dict[str, np.ndarray] = {}
    counter = [0]

    def tname(n: str) -> str:
        if named:
            return n[len("det."):]
        counter[0] += 1
        return str(5000 + counter[0])

    def act() -> str:
        counter[0] += 1
        return f"d{counter[0]}"

    def conv(x: str, name: str, bn: Optional[str], stride: int, k: int, relu: bool, w=None, b=None, tag: str = "") -> str:
        w = raw[name + ".weight"].astype(np.float64) if w is None else w
        if bn is None:
            b = raw[name + ".bias"].astype(np.float64) if b is None else b
        ins = [x]
        wn = tname(name + tag + ".weight")
        if bn is not None and fuse_bn:
            g_, b_, m_, v_ = (raw[f"{bn}.{f_}"].astype(np.float64) for f_ in _BN_FIELDS)
            sc = g_ / np.sqrt(v_ + BN_EPS)
            init[wn] = (w * sc[:, None, None, None]).astype(np.float32)
            bname = tname(name + tag + ".bias")
            init[bname] = (b_ - m_ * sc).astype(np.float32)
            ins += [wn, bname]
        elif bn is None:
            init[wn] = w.astype(np.float32)
            bname = tname(name + tag + ".bias")
            init[bname] = b.astype(np.float32)
            ins += [wn, bname]
        else:
            init[wn] = w.astype(np.float32)
            ins.append(wn)
        y = act()
        nodes.append(Node("Conv", ins, [y], name + tag, {"kernel_shape": [k, k], "strides": [stride, stride], "pads": [k // 2] * 4,
                                                         "dilations": [1, 1], "group": 1}))
        if bn is not None and not fuse_bn:
            g_, b_, m_, v_ = (raw[f"{bn}.{f_}"].astype(np.float64) for f_ in _BN_FIELDS)
            names = []
            for f_, arr in zip(_BN_FIELDS, (g_, b_, m_, v_ + (BN_EPS - eps))):
                tn = tname(f"{bn}.{f_}")
                init[tn] = arr.astype(np.float32)
                names.append(tn)
            z = act()
            nodes.append(Node("BatchNormalization", [y] + names, [z], bn, {"epsilon": float(eps)}))
            y = z
        if relu:
            z = act()
            nodes.append(Node("Relu", [y], [z], name + tag + ".relu"))
            y = z
        return y

    x = conv("data", "det.stem1.conv", "det.stem1.bn", 2, 3, True)
    x = conv(x, "det.stem2.conv", "det.stem2.bn", 2, 3, True)
    feats = {}
    for li, nb in enumerate(blocks, start=1):
        for bi in range(nb):
            p = f"det.layer{li}.{bi}"
            stride = 2 if (bi == 0 and li > 1) else 1
            has_ds = f"{p}.downsample.0.weight" in raw
            ident = x
            if has_ds and shortcut_first:
                ident = conv(x, p + ".downsample.0", p + ".downsample.1", stride, 1, False)
            t = conv(x, p + ".conv1", p + ".bn1", stride, 3, True)
            t = conv(t, p + ".conv2", p + ".bn2", 1, 3, False)
            if has_ds and not shortcut_first:
                ident = conv(x, p + ".downsample.0", p + ".downsample.1", stride, 1, False)
            a = act()
            nodes.append(Node("Add", [t, ident], [a], p + ".add"))
            x = act()
            nodes.append(Node("Relu", [a], [x], p + ".relu"))
        feats[li] = x

    def up2(t: str, tag: str) -> str:
        sn = tname(f"det.fpn.{tag}.scales")
        init[sn] = np.array([1, 1, 2, 2], np.float32)
        y = act()
        nodes.append(Node("Resize", [t, "", sn], [y], f"det.fpn.{tag}", {"mode": "nearest", "coordinate_transformation_mode": "asymmetric",
                                                                         "nearest_mode": "floor"}))
        return y

    def lateral(lv: int) -> str:
        return conv(feats[lv - 1], f"det.fpn.lat{lv}.conv", None, 1, 1, False)

    if laterals_first:
        l3, l4, l5 = lateral(3), lateral(4), lateral(5)
    else:
        l5 = lateral(5)
    p5 = l5
    if not laterals_first:
        l4 = lateral(4)
    p4 = act()
    nodes.append(Node("Add", [l4, up2(p5, "up5")], [p4], "det.fpn.add4"))
    if not laterals_first:
        l3 = lateral(3)
    p3 = act()
    nodes.append(Node("Add", [l3, up2(p4, "up4")], [p3], "det.fpn.add3"))
    outs: List[str] = []
    A, V = ns.DET_NUM_ANCHORS, ns.DET_VALUES_PER_ANCHOR
    for lv, pt in ((3, p3), (4, p4), (5, p5)):
        f = conv(pt, f"det.fpn.smooth{lv}.conv", f"det.fpn.smooth{lv}.bn", 1, 3, True)
        h = f"det.head{lv}"
        f = conv(f, f"{h}.tower0.conv", f"{h}.tower0.bn", 1, 3, True)
        f = conv(f, f"{h}.tower1.conv", f"{h}.tower1.bn", 1, 3, True)
        if not split_heads:
            outs.append(conv(f, f"{h}.out", None, 1, 3, False))
            continue
        w, b = raw[f"{h}.out.weight"].astype(np.float64), raw[f"{h}.out.bias"].astype(np.float64)
        rows = {"score": [a * V for a in range(A)], "bbox": [a * V + 1 + j for a in range(A) for j in range(4)],
                "kps": [a * V + 5 + j for a in range(A) for j in range(10)]}
        for tag, idx in rows.items():
            o = conv(f, f"{h}.out", None, 1, 3, False, w=w[idx], b=b[idx], tag="." + tag)
            if tag == "score" and sigmoid_scores:
                z = act()
                nodes.append(Node("Sigmoid", [o], [z], f"{h}.score.sigmoid"))
                o = z
            outs.append(o)
    return write_model(nodes, init, ["data"], outs, raw_data)
