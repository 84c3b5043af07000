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
# [r6] a SECOND, independent optimiser run (AUTO_TOLERANCE and the 1e-4 claims rested on one training run): other
# initial weights, other torch seed, other clouds, three times the steps, other loss weighting (oracle/make_golden_trained.py, run
# "b").  C = 4 only.  Stored the same way under tests/golden/trained2_*.
RUNS = {"a": {"prefix": "trained", "init_seed": INIT_SEED}, "b": {"prefix": "trained2", "init_seed": 101}}


def weights_path(C: int, run: str = "a") -> str:
    return os.path.join(GOLDEN, f"{RUNS[run]['prefix']}_weights_c{C}.npz")


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


def trained_state_dict(C: int = 4, run: str = "a") -> "OrderedDict[str, torch.Tensor]":
    a4 = np.load(weights_path(4, run))
    sd4 = decode({k: a4[k] for k in a4.files}, synth.synth_state_dict(4, RUNS[run]["init_seed"]))
    if C == 4:
        return sd4
    if run != "a":
        raise ValueError("the second training run has a C = 4 checkpoint only")
    a5 = np.load(weights_path(5))
    sd5 = OrderedDict()
    for k, (shape, _kind) in synth.checkpoint_schema(5).items():
        sd5[k] = torch.from_numpy(np.array(a5[k])) if k in a5.files else sd4[k].clone()
        assert tuple(sd5[k].shape) == tuple(shape), k
    return sd5


def available(run: str = "a") -> bool:
    return os.path.exists(weights_path(4, run)) and (run != "a" or os.path.exists(weights_path(5)))


def run_of_fixture(path: str) -> str:
    """which training run a tests/golden/trained*_*.npz fixture belongs to"""
    return "b" if os.path.basename(path).startswith("trained2_") else "a"
