"""ONNX model-pack reader: wire-format round trips, IResNet and FRPDet mapping, CPU only (the device side:
tests/test_gpu_pipeline.py::test_onnx_packs_reach_the_device)."""
import numpy as np
import pytest

import frp_amd_loader  # noqa: F401
from frp_amd import netspec as ns
from frp_amd import onnx_pack, weights

BLOCKS = (2, 1, 2, 1)


@pytest.fixture(scope="module")
def raw():
    return weights.make_synthetic_raw(11, emb_blocks=BLOCKS, want_det=False)


def _folded(raw, blocks=BLOCKS):
    return [weights.fold_layer(raw, l) for l in ns.iresnet_layers(blocks)]


def test_tensor_encodings_round_trip():
    rng = np.random.default_rng(0)
    tensors = {
        "f32": rng.standard_normal((3, 4, 5)).astype(np.float32),
        "f16": rng.standard_normal((7,)).astype(np.float16),
        "i64": np.array([[-1, 2 ** 40], [0, -(2 ** 35)]], np.int64),
        "scalar": np.float32(3.5).reshape(()),
        "f64": rng.standard_normal((2, 2)),
    }
    nodes = [onnx_pack.Node("Conv", ["x", "f32"], ["y"], "c", {"strides": [2, 2], "group": 1, "alpha": 0.5}),
             onnx_pack.Node("Constant", [], ["k"], "k", {"value": np.arange(4, dtype=np.float32)})]
    for raw_data in (True, False):
        g = onnx_pack.parse_model(onnx_pack.write_model(nodes, tensors, ["x"], ["y"], raw_data))
        for k, v in tensors.items():
            assert g.initializers[k].dtype == v.dtype and g.initializers[k].shape == v.shape
            assert np.array_equal(g.initializers[k], v)
        assert np.array_equal(g.initializers["k"], np.arange(4, dtype=np.float32))      # Constant lifted
        n = g.nodes[0]
        assert (n.op, n.inputs, n.outputs) == ("Conv", ["x", "f32"], ["y"])
        assert n.attrs["strides"] == [2, 2] and n.attrs["group"] == 1 and n.attrs["alpha"] == 0.5
        assert g.inputs == ["x"] and g.outputs == ["y"]


def test_rejects_garbage_and_external_data():
    with pytest.raises(ValueError):
        onnx_pack.parse_model(b"\x0a\x05hello")                 # no graph
    with pytest.raises(ValueError):
        onnx_pack.parse_model(b"\x3a\xff\xff\xff\xff\x0f")      # graph length beyond the buffer
    # TensorProto with an external_data entry (field 13)
    t = onnx_pack._enc(1, 0, 4) + onnx_pack._enc(2, 0, 1) + onnx_pack._enc(8, 2, b"w") + onnx_pack._enc(13, 2, b"\x0a\x01k")
    with pytest.raises(ValueError, match="external data"):
        onnx_pack.parse_model(onnx_pack._enc(7, 2, onnx_pack._enc(5, 2, t)))


@pytest.mark.parametrize("named", [True, False])
def test_unfused_layouts_reproduce_the_raw_dict_exactly(raw, named):
    back = onnx_pack.raw_from_onnx(onnx_pack.iresnet_to_onnx(raw, named=named))
    assert set(back) == {k for k in raw if k.startswith("emb.")}
    for k in back:
        assert np.array_equal(back[k], raw[k]), k
    assert weights.emb_blocks_of(back) == BLOCKS


@pytest.mark.parametrize("kw", [dict(fuse_bn=True), dict(fuse_bn=True, shortcut_first=True, matmul_fc=True, raw_data=False),
                                dict(eps=2e-5), dict(eps=2e-5, shortcut_first=True, matmul_fc=True)])
def test_exporter_variants_fold_to_the_same_layers(raw, kw):
    """BN folded into Conv/Gemm by the exporter, MXNet-style epsilon, shortcut-first node order, MatMul+Add FC:
    the layers the device runs (fp16 weights, fp32 bias / 9-class border bias, slopes) stay the same."""
    back = onnx_pack.raw_from_onnx(onnx_pack.iresnet_to_onnx(raw, named=False, **kw))
    for (w0, b0, s0), (w1, b1, s1), l in zip(_folded(raw), _folded(back), ns.iresnet_layers(BLOCKS)):
        scale = max(1e-3, float(np.abs(w0.astype(np.float32)).max()))
        assert np.abs(w0.astype(np.float32) - w1.astype(np.float32)).max() <= 2e-3 * scale, l.name   # <= 1 fp16 ulp flips
        assert np.allclose(b0, b1, atol=2e-5, rtol=1e-5), l.name
        assert (s0 is None and s1 is None) or np.array_equal(s0, s1)


def test_loaded_pack_gives_the_same_embedding_as_the_source_weights(raw):
    from oracle import network as onet
    back = onnx_pack.raw_from_onnx(onnx_pack.iresnet_to_onnx(raw, named=False, fuse_bn=True, eps=2e-5))
    chips = np.random.default_rng(3).integers(0, 256, (2, 112, 112, 3), dtype=np.uint8)
    e0 = onet.emb_forward(raw, onet.emb_blob(chips))
    e1 = onet.emb_forward(back, onet.emb_blob(chips))
    assert np.abs(e0 - e1).max() < 1e-4
    assert np.sum(e0 * e1, axis=1).min() > 1 - 1e-6


def test_structure_errors_are_reported(raw):
    g = onnx_pack.parse_model(onnx_pack.iresnet_to_onnx(raw, named=False))
    nodes = [n for n in g.nodes if n.op != "PRelu"]            # drop the activations
    with pytest.raises(ValueError, match="expected prelu"):
        onnx_pack.raw_from_onnx(onnx_pack.write_model(nodes, g.initializers, g.inputs, g.outputs))
    bad = dict(g.initializers)
    first_conv = next(n for n in g.nodes if n.op == "Conv")
    bad[first_conv.inputs[1]] = np.zeros((64, 4, 3, 3), np.float32)        # 4 input channels
    with pytest.raises(ValueError, match="emb.conv1.weight"):
        onnx_pack.raw_from_onnx(onnx_pack.write_model(g.nodes, bad, g.inputs, g.outputs))


# ----------------------------------------------------------------------------- detector (FRPDet)
DET_BLOCKS = (1, 2, 1, 1)
DET_LAYOUTS = [dict(), dict(named=False), dict(named=False, fuse_bn=True), dict(named=False, eps=2e-5, shortcut_first=True),
               dict(named=False, laterals_first=True, split_heads=True, sigmoid_scores=True, raw_data=False),
               dict(named=False, fuse_bn=True, split_heads=True, shortcut_first=True, laterals_first=True)]


@pytest.fixture(scope="module")
def det_raw():
    return weights.make_synthetic_raw(13, det_blocks=DET_BLOCKS, want_emb=False)


@pytest.mark.parametrize("kw", DET_LAYOUTS)
def test_detector_layouts_fold_to_the_same_layers(det_raw, kw):
    """named / anonymous initializers, BatchNorms folded by the exporter, MXNet epsilon, shortcut-first and laterals-first node
    order, SCRFD-style split heads (+ Sigmoid on the scores): the reader follows the dataflow, and the layers the device runs
    (fp16 weights, fp32 bias) are the source's - bit for bit where no BatchNorm was folded"""
    back = onnx_pack.det_raw_from_onnx(onnx_pack.detector_to_onnx(det_raw, **kw))
    assert set(back) == set(det_raw) and onnx_pack.det_blocks_of(back) == DET_BLOCKS
    if not kw.get("fuse_bn"):
        for k in det_raw:
            assert np.array_equal(back[k], det_raw[k]), k
    for l in ns.detector_layers(DET_BLOCKS):
        (w0, b0, _), (w1, b1, _) = weights.fold_layer(det_raw, l), weights.fold_layer(back, l)
        scale = max(1e-3, float(np.abs(w0.astype(np.float32)).max()))
        assert np.abs(w0.astype(np.float32) - w1.astype(np.float32)).max() <= 2e-3 * scale, l.name
        assert np.allclose(b0, b1, atol=2e-5, rtol=1e-5), l.name


def test_loaded_detector_gives_the_same_head_maps_as_the_source_weights(det_raw):
    from oracle import network as onet
    back = onnx_pack.det_raw_from_onnx(onnx_pack.detector_to_onnx(det_raw, named=False, fuse_bn=True, split_heads=True, laterals_first=True))
    frames = np.random.default_rng(5).integers(0, 256, (1, 64, 96, 3), dtype=np.uint8)
    x = onet.det_blob(frames, (64, 96))
    for a, b in zip(onet.det_forward(det_raw, x), onet.det_forward(back, x)):
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(a).max())


def test_detector_structure_errors_are_reported(det_raw):
    g = onnx_pack.parse_model(onnx_pack.detector_to_onnx(det_raw, named=False))
    with pytest.raises(ValueError, match="Relu"):
        onnx_pack.det_raw_from_onnx(onnx_pack.write_model([n for n in g.nodes if n.op != "Relu"], g.initializers, g.inputs, g.outputs))
    with pytest.raises(ValueError, match="nearest"):
        nodes = [onnx_pack.Node(n.op, n.inputs, n.outputs, n.name, dict(n.attrs, mode="linear") if n.op == "Resize" else n.attrs) for n in g.nodes]
        onnx_pack.det_raw_from_onnx(onnx_pack.write_model(nodes, g.initializers, g.inputs, g.outputs))
    bad = dict(g.initializers)
    first = next(n for n in g.nodes if n.op == "Conv")
    bad[first.inputs[1]] = np.zeros((32, 4, 3, 3), np.float32)
    with pytest.raises(ValueError, match="det.stem1.conv.weight"):
        onnx_pack.det_raw_from_onnx(onnx_pack.write_model(g.nodes, bad, g.inputs, g.outputs))
    # an embedder file is not a detector, and the other way round
    emb = weights.make_synthetic_raw(11, emb_blocks=(1, 1, 1, 1), want_det=False)
    with pytest.raises(ValueError, match="FRPDet"):
        onnx_pack.det_raw_from_onnx(onnx_pack.iresnet_to_onnx(emb, named=False))
    with pytest.raises(ValueError, match="IResNet"):
        onnx_pack.raw_from_onnx(onnx_pack.detector_to_onnx(det_raw, named=False))


def test_pack_from_onnx_builds_a_loadable_blob(det_raw, tmp_path):
    emb = weights.make_synthetic_raw(11, emb_blocks=(1, 1, 1, 1), want_det=False)
    d, e = tmp_path / "det.onnx", tmp_path / "emb.onnx"
    d.write_bytes(onnx_pack.detector_to_onnx(det_raw, named=False, split_heads=True))
    e.write_bytes(onnx_pack.iresnet_to_onnx(emb, named=False, fuse_bn=True))
    blob = onnx_pack.pack_from_onnx(str(d), str(e))
    both = dict(det_raw)
    both.update(emb)
    ref = weights.pack_blob(both, DET_BLOCKS, (1, 1, 1, 1))
    assert len(blob) == len(ref)                    # same program, same tensor sizes (fused BNs move low bits of the embedder's weights)
