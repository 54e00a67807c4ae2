"""Generate tests/golden/activation_reference.json: outputs of the REFERENCE's own activation functors
(Mila/Src/Dnn/Components/Activations/Activation/Kernels/ElementwiseActivation.h, compiled where it lies into
oracle/_ref/libmila_ref_act.so by oracle/Makefile -- the one reference source that builds in this image) on a fixed input grid.
Inputs and outputs are stored as float32 bit patterns, so the fixture pins the oracle bit for bit on boxes where /root/reference
(and therefore oracle/_ref) is absent.  Run from the repo root in the container that has /root/reference:
    make -C oracle && python tests/golden/make_activation_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402


def inputs():
    rng = np.random.default_rng(20240607)
    xs = np.concatenate([np.linspace(-12, 12, 385), rng.standard_normal(256) * 4, [0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0, 1.0, -2.5]])
    return xs.astype(np.float32)


def main():
    ref = orc.ref_activation_lib()
    if ref is None:
        raise SystemExit("oracle/_ref/libmila_ref_act.so is missing: run `make -C oracle` where /root/reference exists")
    xs = inputs()
    rows = {"x": [], "gelu_tanh": [], "silu": []}
    for x in xs:
        rows["x"].append(int(np.float32(x).view(np.uint32)))
        rows["gelu_tanh"].append(int(np.float32(ref.ref_gelu_tanh(float(x))).view(np.uint32)))
        rows["silu"].append(int(np.float32(ref.ref_silu(float(x))).view(np.uint32)))
    out = {"source": "ElementwiseActivation.h:41-75 (GeluTanh::fwd, Silu::fwd) via oracle/_ref/libmila_ref_act.so",
           "encoding": "float32 bit patterns (uint32)", **rows}
    with open(os.path.join(HERE, "activation_reference.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(xs), "vectors")


if __name__ == "__main__":
    main()
