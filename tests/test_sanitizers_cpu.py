"""ASan + UBSan over the CPU-only pieces (CPU container only -- GPU sanitizers are not available on the pool):
the container readers / writers of Mila/Serialization.h, driven over valid files and a malformed / mutated corpus, and the C oracle
over a battery of small, ragged and degenerate shapes.  tests/cpp/san_driver.cpp is compiled here with
-fsanitize=address,undefined -fno-sanitize-recover=all; a sanitizer report aborts the driver, which fails the test."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    d = tmp_path_factory.mktemp("san")
    obj = d / "oracle.o"
    subprocess.check_call(["gcc", "-std=gnu11", "-ffp-contract=off", "-c", os.path.join(ROOT, "oracle", "mila_oracle.c"), "-o", str(obj)] + SAN)
    exe = d / "san_driver"
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "mila_amd", "host", "include"), "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "san_driver.cpp"), str(obj), "-o", str(exe), "-lm"] + SAN)
    return str(exe)


def run(driver, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([driver, *map(str, args)], capture_output=True, text=True, errors="replace", env=env, timeout=300)
    assert p.returncode in (0, 3), "sanitizer report or crash (rc %d):\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-6000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-6000:]
    return p.returncode, p.stdout


def test_oracle_battery_is_clean_under_asan_ubsan(driver):
    rc, out = run(driver, "oracle")
    assert rc == 0 and "oracle battery done" in out


def _valid_safetensors(path, rng):
    t = {"temb.wte": (rng.integers(0, 65535, (8, 4), dtype=np.uint16), "BF16"), "tf_layer_0.qkv_proj.weight": (rng.integers(0, 255, (6, 8), dtype=np.uint8), "U8"),
         "tf_layer_0.qkv_proj.weight_scale": (rng.standard_normal((6, 2)).astype(np.float32), "F32"), "tf_layer_0.layer_scalar": (np.array([0.75], np.float32), "F32")}
    hdr, blob, off = {"__metadata__": {"mila_quantization": "per_group_fp4_128", "mila_config": json.dumps({"architecture": "gemma4", "vocab_size": 8, "note": 'q " \\ é'})}}, b"", 0
    for k, (a, dt) in t.items():
        b = a.tobytes()
        hdr[k] = {"dtype": dt, "shape": list(a.shape), "data_offsets": [off, off + len(b)]}
        blob += b
        off += len(b)
    text = json.dumps(hdr).encode()
    text += b" " * ((8 - (8 + len(text)) % 8) % 8)
    open(path, "wb").write(struct.pack("<Q", len(text)) + text + blob)


def test_valid_files_round_trip_clean(driver, tmp_path):
    rng = np.random.default_rng(0)
    src = tmp_path / "a.safetensors"
    _valid_safetensors(src, rng)
    assert run(driver, "list", src)[0] == 0
    assert run(driver, "copy", src, tmp_path / "b.safetensors")[0] == 0
    assert run(driver, "tobin", src, tmp_path / "a.bin")[0] == 0
    rc, out = run(driver, "list", tmp_path / "a.bin")
    assert rc == 0 and "container mila" in out and "4 tensors" in out
    assert run(driver, "tobin", tmp_path / "a.bin", tmp_path / "a2.bin")[0] == 0
    assert open(tmp_path / "a.bin", "rb").read() == open(tmp_path / "a2.bin", "rb").read()


def test_mutated_corpus_is_rejected_or_accepted_without_a_sanitizer_report(driver, tmp_path):
    """every single-byte and truncation mutant of the headers of both containers (plus random multi-byte damage): the readers must
    either accept (and then every tensor byte is readable) or throw -- never read out of bounds, overflow, or leak"""
    rng = np.random.default_rng(1)
    src = tmp_path / "a.safetensors"
    _valid_safetensors(src, rng)
    run(driver, "tobin", src, tmp_path / "a.bin")
    outcomes = {0: 0, 3: 0}
    for name in ("a.safetensors", "a.bin"):
        raw = open(tmp_path / name, "rb").read()
        hlen = (8 + struct.unpack("<Q", raw[:8])[0]) if name.endswith("safetensors") else len(raw) - 148     # 148 = the four blobs
        muts = []
        for pos in list(range(0, min(hlen, 96))) + list(rng.integers(0, hlen, 60)):                          # header bytes: low ones densely
            for val in (0x00, 0xFF, raw[pos] ^ 0x01, raw[pos] ^ 0x80):
                muts.append(raw[:pos] + bytes([val & 0xFF]) + raw[pos + 1:])
        muts += [raw[:n] for n in (0, 1, 3, 4, 7, 8, 9, 15, 16, 17, hlen - 1, hlen, hlen + 1, len(raw) - 1)]
        muts += [raw + b"\x00" * 5, raw[:8] + raw[8:hlen][::-1] + raw[hlen:]]
        for _ in range(40):
            b = bytearray(raw)
            for p in rng.integers(0, hlen, 4):
                b[p] = int(rng.integers(0, 256))
            muts.append(bytes(b))
        f = tmp_path / ("mut_" + name)
        for m in muts:
            open(f, "wb").write(m)
            rc, _ = run(driver, "list", f)
            outcomes[rc] += 1
    assert outcomes[3] > 100 and outcomes[0] > 10, outcomes       # most header damage is caught; benign damage (a data byte, a name letter) loads


def test_metadata_parser_on_hostile_text(driver, tmp_path):
    f = tmp_path / "m.json"
    for text in ('', '{', '{"vocab_size"', '{"vocab_size":', '{"vocab_size": }', '{"model_name": "unterminated', '{"model_name": "a\\', '"rope_theta":1e999',
                 '{"vocab_size": 99999999999999999999}', '{"use_bias": true, "use_bias": false}', '{"architecture":"' + "x" * 100000 + '"}', "\x00\xff\xfe" * 50):
        open(f, "wb").write(text.encode("latin-1"))
        assert run(driver, "metadata", f)[0] == 0
