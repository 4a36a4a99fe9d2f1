"""Recipe shared by tools/make_golden.py (which runs the REFERENCE's HandEncoder / LightEstimator / MMPool from source,
network/res_encoder.py:53-209,247-265) and the tests that compare oracle/torch_modules.*Ref and the HIP modules with
tests/golden/heads.npz.  Weights and inputs are NOT stored (the 1536-wide encoder alone is 10 MB): both sides rebuild them from a
numpy RandomState, whose stream is stable across numpy versions; the fixture holds what the reference computed from them."""
import numpy as np
import torch

# name -> (class, constructor args, input shape).  The reference's own configurations: models_res_nimble.py:51-57 (mano [10, 48, None],
# nimble [20, 30, 10]); in_dim 512 is what ResNet-18 needs, 1536 EfficientNet-b3; LightEstimator 512 / 32 as written at :172-175.
CASES = {
    "he_mano512": ("HandEncoder", ("mano", [10, 48, None], 512), (8, 512)),
    "he_nimble1536": ("HandEncoder", ("nimble", [20, 30, 10], 1536), (8, 1536)),
    "le512": ("LightEstimator", (512,), (3, 512, 28, 28)),
    "le32": ("LightEstimator", (32,), (4, 32, 56, 56)),
    "mmpool": ("MMPool", ((1, 1),), (4, 96, 14, 14)),
}
HE_KEYS = ("pose_params", "shape_params", "texture_params", "scale", "trans", "rot")
LE_KEYS = ("colors", "directions")
MAX_FULL = 4096


def _seed(name):
    return 7000 + sum(ord(c) * (i + 1) for i, c in enumerate(name))


def fill_state(module, name):
    """Every parameter and buffer from one numpy stream, in state_dict order: non-zero biases, batch-norm scales around 1,
    running statistics away from (0, 1) -- the reference's own init leaves biases at 0, which would hide a dropped bias."""
    rng = np.random.RandomState(_seed(name))
    sd = module.state_dict()
    new = {}
    for k, t in sd.items():
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            new[k] = torch.zeros_like(t)
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy((0.5 + rng.rand(*shape)).astype(np.float32))
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy((0.2 * rng.standard_normal(shape)).astype(np.float32))
        elif t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            new[k] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        elif k.endswith("weight"):                      # BatchNorm1d scale
            new[k] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif k == "p":                                  # MMPool's mixing scalar
            new[k] = torch.from_numpy(np.full(shape, 0.3, np.float32))
        else:                                           # biases
            new[k] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
    module.load_state_dict(new)
    return module


def make_input(name):
    shape = CASES[name][2]
    rng = np.random.RandomState(_seed(name) + 1)
    x = rng.standard_normal(shape).astype(np.float32)
    if name == "mmpool":                                # exact ties for the max (first-index rule) in a few windows
        x = np.round(x * 4) / 4
    return torch.from_numpy(x)


def projection(name, key, shape):
    rng = np.random.RandomState(_seed(name) + 2 + sum(ord(c) for c in key))
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


def sample(t):
    """What of a tensor goes into the fixture: all of it up to MAX_FULL elements, else an even stride through the flat tensor."""
    a = t.detach().cpu().contiguous().reshape(-1).numpy()
    if a.size <= MAX_FULL:
        return a.copy()
    return a[:: -(-a.size // MAX_FULL)].copy()


def outputs_of(name, out):
    """{key: tensor} of a module's output (None entries dropped), in a fixed order."""
    kind = CASES[name][0]
    if kind == "HandEncoder":
        return {k: out[k] for k in HE_KEYS if out[k] is not None}
    if kind == "LightEstimator":
        return {k: out[k] for k in LE_KEYS}
    return {"y": out}


def run_case(module, name, train, device="cpu"):
    """forward + backward of sum_k <out_k, W_k> on the recipe's weights and input.  Returns (outputs, sampled gradients, buffers)."""
    module = module.to(device)
    module.train(train)
    x = make_input(name).to(device).requires_grad_(True)
    outs = outputs_of(name, module(x))
    loss = sum((o * projection(name, k, o.shape).to(device)).sum() for k, o in outs.items())
    for p in module.parameters():
        p.grad = None
    loss.backward()
    if device != "cpu":
        torch.cuda.synchronize()
    grads = {"x": x.grad}
    for n, p in module.named_parameters():
        if p.grad is not None:
            grads[n] = p.grad
    bufs = {n: b for n, b in module.named_buffers() if "running" in n}
    return outs, grads, bufs
