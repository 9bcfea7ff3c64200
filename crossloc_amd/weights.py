"""Closed-form seeded weights for the scene-coordinate network.

No trained CrossLoc weights exist offline and 107 MB of floats must not be committed, so every
tensor is a pure function of (seed, state_dict key, shape): the golden-vector script fills the
imported reference network with these values and the tests regenerate them for our network.
Scales keep activations O(1) through the 28 GroupNorm layers.
"""
import zlib

import numpy as np
import torch


def seeded_tensor(key, shape, seed=2021):
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) == 4:                                   # conv weight [Cout, Cin, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        a = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
    elif len(shape) == 1 and leaf == "weight":            # GroupNorm gamma
        a = 1.0 + 0.1 * rng.standard_normal(shape)
    elif len(shape) == 1:                                 # conv bias / GroupNorm beta
        a = 0.1 * rng.standard_normal(shape)
    else:
        a = rng.standard_normal(shape)
    return torch.from_numpy(a.astype(np.float32))


def seeded_state_dict(module, seed=2021, skip=("mean",)):
    """state_dict for `module` (ours or the reference's: identical keys) with seeded values;
    buffers named in `skip` (the output-offset `mean`) keep their current values."""
    sd = {}
    for k, v in module.state_dict().items():
        if k.rsplit(".", 1)[-1] in skip:
            sd[k] = v.clone()
        else:
            sd[k] = seeded_tensor(k, v.shape, seed).to(v.dtype)
    return sd
