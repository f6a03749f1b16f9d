"""Real-weight loader for ONNX model packs, without the `onnx` / `protobuf` packages.

The reference gets its networks from downloaded model packs (insightface `FaceAnalysis`,
backend/app/utils/deepfake_utils.py:39-51; dlib .dat files behind face_recognition,
backend/app/services/face_service.py:156,179).  This module reads a user-supplied `.onnx` file
(protobuf wire format parsed by hand), and maps an ArcFace IResNet graph onto the raw dict that
`weights.pack_blob` consumes:

* `parse_model(bytes)`  -> Graph(nodes, initializers)            (generic, any ONNX file)
* `raw_from_onnx(path)` -> {"emb.conv1.weight": ..., ...}        (IResNet-50/100 family)
* `write_model(...)`    -> bytes                                  (minimal writer, used by
  tools/export_onnx.py and the tests to produce files in the layouts exporters emit)

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
