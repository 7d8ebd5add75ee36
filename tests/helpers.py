"""Shared helpers for the parity tests (fixtures + deterministic weights)."""
import json
import os

import numpy as np
import torch

from oracle import weights as oweights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if k == "cfg":
            out[k] = json.loads(str(v))
        elif v.dtype.kind == "f" and v.ndim > 0:
            out[k] = torch.from_numpy(v.copy())
        else:
            out[k] = v
    return out


def weight_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def filled(template, prefix=""):
    """Fill a template state dict by name; ``prefix`` is prepended to each key for seeding only."""
    out = {}
    for k in sorted(template):
        v = oweights.fill_value(prefix + k, template[k])
        out[k] = v.to(template[k].dtype).reshape(template[k].shape).contiguous()
    return out


def maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())
