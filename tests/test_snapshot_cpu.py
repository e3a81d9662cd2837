"""CPU tests of the snapshot container: the MessagePack codec of `.msgpack` snapshots (reference: json::to_msgpack / from_msgpack,
src/testbed.cu:3041, 139) against the independent `msgpack` Python package, and the host fp16 conversion against numpy."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

msgpack = pytest.importorskip("msgpack")


@pytest.fixture(scope="module")
def pyngp():
    import torch  # noqa: F401  (one HIP runtime per process: torch first)
    import pyngp as m
    return m


def _sample_tree():
    rs = np.random.RandomState(3)
    return {
        "encoding": {"otype": "HashGrid", "n_levels": 16, "log2_hashmap_size": 19, "per_level_scale": 1.3819128274917603},
        "ints": [0, 1, 127, 128, 255, 256, 65535, 65536, 2**32 - 1, 2**32, 2**40, -1, -32, -33, -128, -129, -32768, -32769, -2**31, -2**31 - 1],
        "floats": [0.5, 1.0e-2, 0.1, 3.0e38, 1e300, -2.25],
        "none": None, "t": True, "f": False,
        "short": "abc", "s31": "x" * 31, "s32": "y" * 32, "s300": "z" * 300, "s70000": "w" * 70000,
        "bin_small": bytes(range(10)), "bin_300": rs.bytes(300), "bin_70000": rs.bytes(70000),
        "arr16": list(range(16)), "arr70000": [1] * 70000,
        "map20": {"k%02d" % i: i for i in range(20)},
        "nested": {"snapshot": {"version": 1, "aabb": {"min": [0.0, 0.0, 0.0], "max": [1.0, 1.0, 1.0]}}},
    }


def test_written_msgpack_is_read_by_python_msgpack(pyngp):
    tree = _sample_tree()
    raw = pyngp.json_to_msgpack(tree)
    back = msgpack.unpackb(raw, raw=False, strict_map_key=True)
    assert back == tree


def test_python_msgpack_is_read_back(pyngp):
    tree = _sample_tree()
    raw = msgpack.packb(tree, use_bin_type=True)
    assert pyngp.msgpack_to_json(raw) == tree
    raw32 = msgpack.packb({"a": [0.5, 0.1]}, use_bin_type=True, use_single_float=True)
    got = pyngp.msgpack_to_json(raw32)
    assert got == {"a": [0.5, float(np.float32(0.1))]}


def test_encodings_are_minimal_and_sorted(pyngp):
    # without floats the two writers must agree byte for byte (smallest int / str / bin / array / map formats, keys in sorted order)
    tree = {k: v for k, v in _sample_tree().items() if k not in ("floats", "encoding", "nested")}
    ordered = {k: (dict(sorted(v.items())) if isinstance(v, dict) else v) for k, v in sorted(tree.items())}
    assert pyngp.json_to_msgpack(tree) == msgpack.packb(ordered, use_bin_type=True)
    # float32 when lossless, float64 otherwise (nlohmann's rule)
    assert pyngp.json_to_msgpack(0.5) == b"\xca" + struct.pack(">f", 0.5)
    assert pyngp.json_to_msgpack(0.1) == b"\xcb" + struct.pack(">d", 0.1)


def test_truncated_and_foreign_input_is_rejected(pyngp):
    raw = pyngp.json_to_msgpack({"a": [1, 2, 3], "b": b"1234"})
    with pytest.raises(RuntimeError):
        pyngp.msgpack_to_json(raw[:-1])
    with pytest.raises(RuntimeError):
        pyngp.msgpack_to_json(raw + b"\x00")
    with pytest.raises(RuntimeError):
        pyngp.msgpack_to_json(b"\xc7\x01\x05\x00")   # ext8
    with pytest.raises(RuntimeError):
        pyngp.msgpack_to_json(msgpack.packb({1: 2}))  # non-string key


def test_half_conversion_matches_numpy(pyngp):
    rs = np.random.RandomState(0)
    vals = np.concatenate([
        rs.uniform(-70000, 70000, 20000), rs.uniform(-1, 1, 20000) * 10.0 ** rs.uniform(-9, 0, 20000),
        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, np.inf, -np.inf, 5.9604645e-08, 2.9802322e-08, 2.98023224e-08 * 1.0001, 6.1035156e-05, 6.0975552e-05],
        np.float16(rs.uniform(-4, 4, 2000)).astype(np.float64) + 2.0 ** -13,   # near ties
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = pyngp.float_to_half_bits(vals)
    np.testing.assert_array_equal(got, want)
    allh = np.arange(65536, dtype=np.uint16)
    f = pyngp.half_bits_to_float(allh)
    ref = allh.view(np.float16).astype(np.float32)
    np.testing.assert_array_equal(f.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    assert np.isnan(f[np.isnan(ref)]).all()
