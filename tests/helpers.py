"""Shared helpers for the test-suite (fixture loading)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def split_fixture(z):
    """-> (state_dict tensors, grads tensors, other arrays)"""
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in z.items() if k.startswith("sd.")}
    gr = {k[5:]: torch.from_numpy(np.array(v)) for k, v in z.items() if k.startswith("grad.")}
    rest = {k: v for k, v in z.items() if not k.startswith(("sd.", "grad."))}
    return sd, gr, rest


def t(a):
    return torch.from_numpy(np.array(a))


def cfg_of(z):
    c = json.loads(str(z["cfg"]))
    for k in ("conv_dim", "conv_kernel", "conv_stride"):
        if k in c:
            c[k] = tuple(c[k])
    return c
