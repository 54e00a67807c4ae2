"""Host restatement of the device-side synthetic parameter generator
(mila_amd/csrc/elementwise.hip: synth_uniform / fill_uniform_bf16_kernel)."""
import numpy as np

import orc

M64 = (1 << 64) - 1


def uniform(seed, n):
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & M64) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def fill_bf16(seed, n, amp, offset):
    """bf16 bit patterns of offset + amp * (2u - 1), computed in fp32 exactly like the kernel"""
    u = uniform(seed, n)
    v = np.float32(offset) + np.float32(amp) * (np.float32(2.0) * u - np.float32(1.0))
    return orc.to_bf16_bits(v.astype(np.float32))
