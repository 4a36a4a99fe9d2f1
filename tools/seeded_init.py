"""Name-keyed deterministic weights (numpy RandomState seeded by crc32 of the parameter name): lets a golden
fixture pin an 11 M-parameter network without storing its weights.  Used by tools/make_golden.py (on the
reference's vendored ResNet) and by tests (on this package's ResNet18Trunk, which uses the same names)."""
import zlib

import numpy as np
import torch


def seeded_state_dict(module: torch.nn.Module) -> dict:
    out = {}
    for name, t in module.state_dict().items():
        rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros_like(t)
        elif name.endswith("running_var"):
            out[name] = torch.ones_like(t)
        elif t.dim() == 4:                                   # conv weight: He init on fan_out
            fan_out = t.shape[0] * t.shape[2] * t.shape[3]
            out[name] = torch.from_numpy(rs.randn(*t.shape).astype(np.float32) * np.sqrt(2.0 / fan_out))
        elif t.dim() == 2:
            out[name] = torch.from_numpy(rs.randn(*t.shape).astype(np.float32) * np.sqrt(1.0 / t.shape[1]))
        elif name.endswith("weight"):                        # BN gamma
            out[name] = torch.from_numpy(1.0 + 0.1 * rs.randn(*t.shape).astype(np.float32))
        else:                                                # biases / BN beta
            out[name] = torch.from_numpy(0.1 * rs.randn(*t.shape).astype(np.float32))
    return out
