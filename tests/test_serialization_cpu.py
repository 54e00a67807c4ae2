"""SafeTensors container of the host mirror (mila_amd/host/include/Mila/Serialization.h) against the
Python `safetensors` package: files written by either side load in the other, the C++ reader rejects malformed containers.
Host-only: no device call."""
import json
import os
import struct

import numpy as np
import pytest

from mila_amd import host

st_numpy = pytest.importorskip("safetensors.numpy")


def _tensors(rng):
    return {"gemma.layer_0.qkv_proj.weight": rng.integers(0, 255, (6, 8), dtype=np.uint8),
            "gemma.layer_0.qkv_proj.weight_scale": rng.standard_normal((6, 2)).astype(np.float32),
            "gemma.layer_0.input_norm.weight": rng.integers(0, 65535, (16,), dtype=np.uint16),
            "gemma.layer_0.layer_scalar": np.array([0.75], dtype=np.float32),
            "empty": np.zeros((0, 4), dtype=np.float32)}


def test_reader_lists_what_the_python_package_wrote(tmp_path):
    rng = np.random.default_rng(0)
    t = _tensors(rng)
    p = tmp_path / "a.safetensors"
    st_numpy.save_file(t, str(p), metadata={"mila_quantization": "PerGroupFp4<128>", "note": 'quote " and \\ backslash'})
    got, meta = host.safetensors_list(p)
    assert meta == {"mila_quantization": "PerGroupFp4<128>", "note": 'quote " and \\ backslash'}
    names = {n: (d, b, s) for n, d, b, s in got}
    assert set(names) == set(t)
    assert names["gemma.layer_0.qkv_proj.weight"] == ("U8", 48, (6, 8))
    assert names["gemma.layer_0.qkv_proj.weight_scale"] == ("F32", 48, (6, 2))
    assert names["gemma.layer_0.input_norm.weight"] == ("U16", 32, (16,))
    assert names["empty"] == ("F32", 0, (0, 4))


def test_writer_output_loads_in_the_python_package_bit_for_bit(tmp_path):
    rng = np.random.default_rng(1)
    t = _tensors(rng)
    src, dst = tmp_path / "src.safetensors", tmp_path / "dst.safetensors"
    st_numpy.save_file(t, str(src), metadata={"format": "pt", "mila_config": json.dumps({"num_layers": 1})})
    host.safetensors_copy(src, dst)                          # C++ reader -> C++ writer
    back = st_numpy.load_file(str(dst))
    assert set(back) == set(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape and np.array_equal(back[k], t[k]), k
    raw = open(dst, "rb").read()
    hlen = struct.unpack("<Q", raw[:8])[0]
    assert (8 + hlen) % 8 == 0                               # header padded to 8 bytes, as the reference writer does
    hdr = json.loads(raw[8:8 + hlen])
    assert hdr["__metadata__"] == {"format": "pt", "mila_config": json.dumps({"num_layers": 1})}
    offs = sorted(v["data_offsets"] for k, v in hdr.items() if k != "__metadata__")
    assert offs[0][0] == 0 and all(a[1] == b[0] for a, b in zip(offs, offs[1:])) and offs[-1][1] == len(raw) - 8 - hlen


@pytest.mark.parametrize("damage", ["truncated", "bad_offsets", "shape_mismatch", "not_json", "missing"])
def test_reader_rejects_malformed_containers(tmp_path, damage):
    p = tmp_path / "x.safetensors"
    hdr = {"w": {"dtype": "F32", "shape": [2, 2], "data_offsets": [0, 16]}}
    data = b"\x00" * 16
    if damage == "bad_offsets":
        hdr["w"]["data_offsets"] = [0, 32]
    elif damage == "shape_mismatch":
        hdr["w"]["shape"] = [3, 2]
    text = json.dumps(hdr).encode()
    if damage == "not_json":
        text = b"{w: oops}"
    blob = struct.pack("<Q", len(text)) + text + data
    if damage == "truncated":
        blob = struct.pack("<Q", len(text) + 1000) + text
    if damage != "missing":
        open(p, "wb").write(blob)
    with pytest.raises((ValueError, RuntimeError)):
        host.safetensors_list(p)
    if damage == "missing":
        assert not os.path.exists(p)
