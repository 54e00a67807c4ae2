"""Weight containers of the host mirror (mila_amd/host/include/Mila/Serialization.h).
SafeTensors against the Python `safetensors` package: files written by either side load in the other.  The reference's MILA .bin
container (Serialization/PretrainedReader.ixx:222-231, :1121-1306) against an independent Python writer / parser of the byte layout
that file describes.  The C++ readers reject malformed containers of both kinds.  Host-only: no device call."""
import json
import os
import struct

import numpy as np
import pytest

from mila_amd import host

st_numpy = pytest.importorskip("safetensors.numpy")


def _tensors(rng):
    return {"tf_layer_0.qkv_proj.weight": rng.integers(0, 255, (6, 8), dtype=np.uint8),
            "tf_layer_0.qkv_proj.weight_scale": rng.standard_normal((6, 2)).astype(np.float32),
            "tf_layer_0.input_norm.weight": rng.integers(0, 65535, (16,), dtype=np.uint16),
            "tf_layer_0.layer_scalar": np.array([0.75], dtype=np.float32),
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
    assert names["tf_layer_0.qkv_proj.weight"] == ("U8", 48, (6, 8))
    assert names["tf_layer_0.qkv_proj.weight_scale"] == ("F32", 48, (6, 2))
    assert names["tf_layer_0.input_norm.weight"] == ("U16", 32, (16,))
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


# ---- MILA .bin ------------------------------------------------------------------------------------------------------------------
MILA_MAGIC, MILA_VERSION = 0x4D494C41, 1
WIRE = {"F32": 0, "F16": 1, "BF16": 2, "I32": 3, "U8": 4, "F8_E4M3": 5, "F8_E5M2": 6, "I8": 7}       # PretrainedReader.ixx:182-191
META = {"architecture": "gemma4", "model_name": "rope_theta", "vocab_size": 262144, "max_seq_length": 4096, "embedding_dim": 3840,
        "num_layers": 48, "num_heads": 16, "num_kv_heads": 8, "head_dim": 256, "hidden_dim": 15360, "use_bias": False,
        "tie_word_embeddings": True, "activation": "gelu", "norm_type": "rmsnorm", "attention_type": "gqa",
        "positional_encoding": "rope", "rope_theta": 10000.0, "norm_epsilon": 1e-6, "global_head_dim": 512,
        "num_global_kv_heads": 1, "key_equals_value": True, "window": 1024, "sliding_window_pattern": 6, "global_rotary_dim": 128,
        "rope_theta_local": 10000.0, "rope_theta_global": 1000000.0, "final_logit_softcapping": 30.0}


def write_mila_bin(path, tensors, meta_json, order=None, patch=None):
    """independent writer of the layout: header | metadata | index (absolute offsets) | blobs.  tensors: [(name, dtype, shape, bytes)];
    `order` permutes where the blobs land in the file (the index keeps declaration order); `patch(index_records)` may damage a record."""
    recs = [dict(name=n.encode(), dtype=WIRE[d], shape=list(sh), nbytes=len(b), blob=b) for n, d, sh, b in tensors]
    index_bytes = sum(4 + len(r["name"]) + 4 + 4 + 4 * len(r["shape"]) + 8 + 8 for r in recs)
    off = 16 + len(meta_json) + index_bytes
    place = list(range(len(recs))) if order is None else list(order)
    for i in place:
        recs[i]["offset"] = off
        off += recs[i]["nbytes"]
    if patch:
        patch(recs)
    out = struct.pack("<IIII", MILA_MAGIC, MILA_VERSION, len(recs), len(meta_json)) + meta_json
    for r in recs:
        out += struct.pack("<I", r.get("name_len", len(r["name"]))) + r["name"] + struct.pack("<II", r["dtype"], r.get("rank", len(r["shape"])))
        out += b"".join(struct.pack("<I", d) for d in r["shape"]) + struct.pack("<QQ", r["offset"], r["nbytes"])
    for i in place:
        out += recs[i]["blob"]
    open(path, "wb").write(out)
    return recs


def parse_mila_bin(path):
    raw = open(path, "rb").read()
    magic, version, n, mlen = struct.unpack_from("<IIII", raw, 0)
    assert magic == MILA_MAGIC and version == MILA_VERSION
    pos = 16
    meta = raw[pos:pos + mlen].decode()
    pos += mlen
    out = {}
    for _ in range(n):
        (nl,) = struct.unpack_from("<I", raw, pos); pos += 4
        name = raw[pos:pos + nl].decode(); pos += nl
        dtype, rank = struct.unpack_from("<II", raw, pos); pos += 8
        shape = struct.unpack_from("<%dI" % rank, raw, pos); pos += 4 * rank
        off, nb = struct.unpack_from("<QQ", raw, pos); pos += 16
        out[name] = (dtype, tuple(shape), raw[off:off + nb], off)
    return meta, out, pos


def _bin_tensors(rng):
    return [("temb.wte", "BF16", (8, 4), rng.integers(0, 255, 64, dtype=np.uint8).tobytes()),
            ("tf_layer_0.qkv_proj.weight", "U8", (6, 8), rng.integers(0, 255, 48, dtype=np.uint8).tobytes()),
            ("tf_layer_0.qkv_proj.weight_scale", "F32", (6, 2), rng.standard_normal(12).astype(np.float32).tobytes()),
            ("tf_layer_0.layer_scalar", "F32", (1,), np.array([0.75], np.float32).tobytes()),
            ("rmsn_final.weight", "BF16", (4,), rng.integers(0, 255, 8, dtype=np.uint8).tobytes())]


def test_mila_bin_reader_lists_an_independently_written_container_in_offset_order(tmp_path):
    rng = np.random.default_rng(3)
    t = _bin_tensors(rng)
    p = tmp_path / "m.bin"
    recs = write_mila_bin(p, t, json.dumps(META).encode(), order=[3, 0, 4, 2, 1])       # blobs land in another order than the index
    got, meta = host.pretrained_list(p)
    assert meta["container"] == "mila" and meta["mila_quantization"] == ""
    assert json.loads(meta["mila_config"]) == META                                        # the block is carried verbatim
    by_off = sorted(recs, key=lambda r: r["offset"])
    assert [g[0] for g in got] == [r["name"].decode() for r in by_off]                    # ascending file offsets: one sequential pass
    want = {n: (d, len(b), tuple(sh)) for n, d, sh, b in t}
    assert {g[0]: g[1:] for g in got} == want


def test_safetensors_to_mila_bin_keeps_every_byte(tmp_path):
    rng = np.random.default_rng(4)
    t = _tensors(rng)
    t.pop("empty")
    u16 = t.pop("tf_layer_0.input_norm.weight")             # numpy has no bf16; U16 has no MILA wire code (checked below)
    t["tf_layer_0.pos"] = rng.integers(-5, 5, (3,), dtype=np.int32)
    src, dst = tmp_path / "a.safetensors", tmp_path / "a.bin"
    st_numpy.save_file(t, str(src), metadata={"mila_config": json.dumps(META), "mila_quantization": "per_group_fp4_128"})
    listed, meta = host.pretrained_list(src)
    assert meta["container"] == "safetensors" and meta["mila_quantization"] == "per_group_fp4_128" and json.loads(meta["mila_config"]) == META
    host.pretrained_to_milabin(src, dst)
    mj, tensors, data0 = parse_mila_bin(dst)
    assert json.loads(mj) == META
    assert set(tensors) == set(t)
    for k, a in t.items():
        dtype, shape, blob, off = tensors[k]
        assert shape == a.shape and blob == a.tobytes() and off >= data0, k
    assert tensors["tf_layer_0.pos"][0] == WIRE["I32"] and tensors["tf_layer_0.qkv_proj.weight"][0] == WIRE["U8"]
    again, meta2 = host.pretrained_list(dst)
    assert meta2["container"] == "mila" and {g[0]: g[2] for g in again} == {g[0]: g[2] for g in listed}
    st_numpy.save_file({"w": u16}, str(src))
    with pytest.raises(RuntimeError, match="no wire code"):     # the container's dtype set is closed (PretrainedReader.ixx:209-220)
        host.pretrained_to_milabin(src, dst, "{}")


def test_metadata_parser_matches_keys_not_substrings_or_values():
    """'rope_theta' must not match inside 'rope_theta_local' nor a string VALUE that spells a key (model_name = "rope_theta")"""
    def f32(m):       # the metadata struct holds floats as FP32, like the reference's
        return {k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in m.items()}
    back = json.loads(host.pretrained_metadata_roundtrip(json.dumps(META)))
    assert f32(back) == f32(META)
    shuffled = dict(reversed(list(META.items())))                                        # key order must not matter
    assert f32(json.loads(host.pretrained_metadata_roundtrip(json.dumps(shuffled, indent=2)))) == f32(META)
    assert json.loads(host.pretrained_metadata_roundtrip("{}"))["vocab_size"] == 0        # absent keys: the reference parser's defaults


@pytest.mark.parametrize("damage", ["truncated_header", "truncated_index", "bad_version", "zero_name", "long_name", "rank", "past_eof",
                                    "shape_bytes", "overlap", "duplicate", "huge_count", "dtype", "into_header", "no_metadata",
                                    "shape_overflow"])
def test_mila_bin_reader_rejects_malformed_containers(tmp_path, damage):
    rng = np.random.default_rng(5)
    t = _bin_tensors(rng)
    p = tmp_path / "bad.bin"
    mj = json.dumps(META).encode()

    def patch(recs):
        if damage == "zero_name":
            recs[1]["name_len"] = 0
        elif damage == "long_name":
            recs[1]["name_len"] = 5000
        elif damage == "rank":
            recs[2]["rank"] = 9
        elif damage == "past_eof":
            recs[4]["offset"] += 1 << 20
        elif damage == "shape_bytes":
            recs[0]["shape"] = [8, 5]
        elif damage == "overlap":
            recs[1]["offset"] = recs[0]["offset"] + 4
        elif damage == "duplicate":
            recs[4]["name"] = recs[3]["name"]
        elif damage == "dtype":
            recs[0]["dtype"] = 99
        elif damage == "into_header":
            recs[3]["offset"] = 8
        elif damage == "shape_overflow":
            recs[0]["shape"] = [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF]
    write_mila_bin(p, t, b"" if damage == "no_metadata" else mj, patch=patch)
    raw = open(p, "rb").read()
    if damage == "truncated_header":
        raw = raw[:10]
    elif damage == "truncated_index":
        raw = raw[:16 + len(mj) + 20]
    elif damage == "bad_version":
        raw = raw[:4] + struct.pack("<I", 7) + raw[8:]
    elif damage == "huge_count":
        raw = raw[:8] + struct.pack("<I", 0xFFFFFFF0) + raw[12:]
    open(p, "wb").write(raw)
    with pytest.raises((ValueError, RuntimeError)):
        host.pretrained_list(p)


# ------------------------------------------------------------------------------------------------------------------------------
# The reference's own 16 container scenarios (Tests/Dnn/Serialization/SafeTensors.Cpu.cpp:133-532), one to one against the host
# mirror's containers: tests/cpp/safetensors_scenarios.cpp holds them under the reference's test names (same call sequences, values,
# expectations, and the exception type its tests expect, std::runtime_error); each runs as its own process.
# ------------------------------------------------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAFETENSORS_CPU_CPP = [(133, "RoundTripsTensorsOfMixedDataTypes"), (187, "CarriesMilaConfigThroughMetadata"), (220, "ReadsAFileThatCarriesNoMilaConfig"),
                       (242, "MetadataSurvivesAFullWriteReadCycle"), (320, "SurfacesTheDeclaredWeightQuantization"), (340, "TreatsAnUnquantizedDeclarationAsAbsent"),
                       (368, "LegacyMilaContainerDeclaresNoQuantization"), (384, "RejectsOutOfOrderBodyWrites"), (402, "RejectsBodySizeMismatch"),
                       (417, "RejectsDuplicateTensorNames"), (429, "RejectsDeclarationAfterHeaderIsWritten"), (442, "CloseRefusesWhenADeclaredTensorWasNeverWritten"),
                       (463, "RejectsAFileThatIsNeitherContainer"), (478, "RejectsATensorExtendingPastEndOfFile"), (502, "StillReadsTheLegacyMilaContainer"),
                       (532, "LegacyContainerStillRejectsAWrongVersion")]


@pytest.fixture(scope="module")
def scenario_driver(tmp_path_factory):
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = tmp_path_factory.mktemp("st_scenarios") / "safetensors_scenarios"
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "mila_amd", "host", "include"),
                           os.path.join(ROOT, "tests", "cpp", "safetensors_scenarios.cpp"), "-o", str(exe)])
    listed = subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()
    assert sorted(listed) == sorted(n for _, n in SAFETENSORS_CPU_CPP)          # the driver holds exactly the reference's 16
    return str(exe)


@pytest.mark.parametrize("line,name", SAFETENSORS_CPU_CPP, ids=["SafeTensors_Cpu_cpp_%d_%s" % ln for ln in SAFETENSORS_CPU_CPP])
def test_reference_container_scenario(scenario_driver, tmp_path, line, name):
    import subprocess
    p = subprocess.run([scenario_driver, name, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, "SafeTensors.Cpu.cpp:%d %s\n%s%s" % (line, name, p.stdout, p.stderr)
