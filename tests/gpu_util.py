"""Helpers shared by the GPU parity tests: device buffers as torch tensors, bf16 bit handling,
ulp-aware comparison against the float64 oracle."""
import ctypes as C

import numpy as np
import torch

from mila_amd import capi

DEV = "cuda:0"


def dev_u16(bits):
    """uint16 numpy (bf16 bit patterns) -> int16 torch tensor on the GPU."""
    return torch.from_numpy(np.ascontiguousarray(bits, dtype=np.uint16).view(np.int16)).to(DEV)


def dev_u8(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(DEV)


def dev_f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def dev_i32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)


def empty_u16(*shape):
    return torch.empty(shape, dtype=torch.int16, device=DEV)


def empty_f32(*shape):
    return torch.empty(shape, dtype=torch.float32, device=DEV)


def empty_u8(*shape):
    return torch.empty(shape, dtype=torch.uint8, device=DEV)


def bits(t):
    """int16 device tensor -> uint16 numpy bit patterns."""
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint16)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def bf16_bits_to_f32(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    """RNE, numpy only (the oracle's converter is checked against this in test_oracle_kats)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def _ordered(bits16):
    b = np.asarray(bits16, dtype=np.uint16).astype(np.int32)
    return np.where(b & 0x8000, -(b & 0x7fff), b)


def assert_bf16_close(got_bits, expected, max_ulp=1, atol=0.0, what=""):
    """got (bf16 bits) must be within `max_ulp` bf16 ulps of RNE(expected) or within atol."""
    got_bits = np.asarray(got_bits).reshape(-1)
    expected = np.asarray(expected, dtype=np.float64).reshape(-1)
    assert got_bits.size == expected.size, (got_bits.size, expected.size)
    exp_bits = f32_to_bf16_bits(expected.astype(np.float32))
    ulp = np.abs(_ordered(got_bits) - _ordered(exp_bits))
    got = bf16_bits_to_f32(got_bits).astype(np.float64)
    bad = (ulp > max_ulp) & ~(np.abs(got - expected) <= atol)
    bad |= ~np.isfinite(got)
    if bad.any():
        i = int(np.argmax(bad))
        raise AssertionError("%s: %d/%d elements off by more than %d bf16 ulp (atol %g); first at %d: got %r exp %r (%d ulp)"
                             % (what, int(bad.sum()), bad.size, max_ulp, atol, i, got[i], expected[i], int(ulp[i])))
    return int(ulp.max()) if ulp.size else 0


def rel_err(got, expected):
    got = np.asarray(got, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    return np.abs(got - expected).max() / max(np.abs(expected).max(), 1e-30)


def call(name, *args):
    capi.call(name, *args)


def size_t(v):
    return C.c_size_t(v)


def i64(v):
    return C.c_int64(v)
