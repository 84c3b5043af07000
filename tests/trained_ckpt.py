"""Test infrastructure: the TRAINED checkpoints of tests/golden/trained_weights_c{4,5}.npz.

oracle/make_golden_trained.py ran the reference's own TEHNet (imported from /root/reference in the build container) in train
mode under Adam (lr 1e-3, /root/reference/src/Ev2Hands/train.py:22-23,53) starting from `synth_state_dict(4, INIT_SEED)`.
What is committed is not the weights but what the optimiser did to them:

  * every conv / linear weight matrix with >= 1024 elements as a float16 DELTA on the hash-generated initial value
    (`<key>::d16`; the checkpoint IS `init + float32(delta)`, evaluated here exactly as the generator evaluated it
    before it ran the reference's forward on it -- so the fixtures are outputs of the reference on exactly these
    float32 weights);
  * everything else (biases, BN affine parameters, BN running statistics, `num_batches_tracked`) verbatim.

`trained_weights_c5.npz` holds only the entries of the C = 5 checkpoint that differ from the C = 4 one: enc.sa1's three
first convolutions (fan-in 8 instead of 7) and their BatchNorms, re-trained with the rest frozen (see the generator).
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from ev2hands_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INIT_SEED = 100
DELTA_MIN_NUMEL = 1024


def weights_path(C: int) -> str:
    return os.path.join(GOLDEN, f"trained_weights_c{C}.npz")


def encode(trained: "OrderedDict[str, torch.Tensor]", init: "OrderedDict[str, torch.Tensor]") -> dict:
    """state dict -> arrays to store (generator side)."""
    out = {}
    for k, v in trained.items():
        if v.dtype == torch.float32 and v.dim() >= 2 and v.numel() >= DELTA_MIN_NUMEL and k in init and init[k].shape == v.shape:
            out[k + "::d16"] = (v - init[k]).numpy().astype(np.float16)
        else:
            out[k] = v.numpy()
    return out


def decode(arrays, init: "OrderedDict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for k in init.keys():
        if k + "::d16" in arrays:
            sd[k] = init[k] + torch.from_numpy(arrays[k + "::d16"].astype(np.float32))
        elif k in arrays:
            sd[k] = torch.from_numpy(np.array(arrays[k]))
        else:
            raise KeyError(k)
    return sd


def trained_state_dict(C: int = 4) -> "OrderedDict[str, torch.Tensor]":
    a4 = np.load(weights_path(4))
    sd4 = decode({k: a4[k] for k in a4.files}, synth.synth_state_dict(4, INIT_SEED))
    if C == 4:
        return sd4
    a5 = np.load(weights_path(5))
    sd5 = OrderedDict()
    for k, (shape, _kind) in synth.checkpoint_schema(5).items():
        sd5[k] = torch.from_numpy(np.array(a5[k])) if k in a5.files else sd4[k].clone()
        assert tuple(sd5[k].shape) == tuple(shape), k
    return sd5


def available() -> bool:
    return os.path.exists(weights_path(4)) and os.path.exists(weights_path(5))
