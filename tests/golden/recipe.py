"""Closed-form, platform-independent recipe for weights and inputs.

Golden vectors (tests/golden/*.npz) store only OUTPUTS of the reference; the
inputs and the state_dict are regenerated from this recipe on every machine
(the reference itself never travels, and MFB's weights are 240 MB).  Every
value comes from an integer hash (splitmix64) of the element index, so the
tensors are bit-identical on any IEEE-754 platform and any numpy version --
no libm, no torch RNG stream.

This file is DATA PLUMBING for tests and bench; it contains no model math.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def unit_uniform(numel, seed):
    """numel values in [0,1), each a multiple of 2^-24 (exact in fp32)."""
    idx = np.arange(numel, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = idx + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x100000001B3)
    h = _splitmix64(key)
    return (h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def name_seed(name, salt=0):
    return (zlib.crc32(name.encode()) + 7919 * salt) & 0xFFFFFFFF


def sym_tensor(shape, amp, seed):
    """fp32 tensor uniform in [-amp, amp)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit_uniform(n, seed)
    return ((u - 0.5) * (2.0 * amp)).astype(np.float32).reshape(shape)


def weight_for(name, shape, salt=0):
    """Recipe value for one state_dict entry, keyed by name and shape.

    Non-bias tensors get the xavier-uniform bound of their shape (what
    train_models.py:54-56 applies), biases a small fixed amplitude.
    """
    seed = name_seed(name, salt)
    shape = tuple(int(s) for s in shape)
    if name.endswith("bias") or "bias" in name.split(".")[-1]:
        return sym_tensor(shape, 0.05, seed)
    if len(shape) < 2:
        return sym_tensor(shape, 0.05, seed)
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_out, fan_in = shape[0] * rf, shape[1] * rf
    bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    # embeddings / tiny fan sums: keep activations O(1)
    bound = min(bound, 0.5)
    # a touch above xavier so that attention logits are not degenerate
    return sym_tensor(shape, 1.5 * bound, seed)


def make_state_dict(shapes, salt=0):
    """shapes: {name: shape}.  Returns {name: np.float32 array}."""
    return {k: weight_for(k, v, salt) for k, v in shapes.items()}


def img_features(n, l, d, salt=0):
    """(n, l, d) fp32, non-negative and ~45 % zeros like post-ReLU CNN grids."""
    x = sym_tensor((n, l, d), 1.0, name_seed("img_features", salt))
    return np.maximum(x - 0.1, 0.0).astype(np.float32)


def question_tokens(n, t, vocab, salt=0, pad_tail=True):
    """(n, t) int64 in [1, vocab); optional zero padding of ragged tails."""
    u = unit_uniform(n * t, name_seed("questions", salt)).reshape(n, t)
    tok = 1 + np.floor(u * (vocab - 1)).astype(np.int64)
    if pad_tail:
        lens = 1 + (np.arange(n) * 5 + 3) % t          # ragged lengths in [1,t]
        lens[0] = t
        for i in range(n):
            tok[i, lens[i]:] = 0
    return tok


def question_lengths(tok):
    """number of non-pad tokens per row (>=1)."""
    return np.maximum((tok != 0).sum(1), 1).astype(np.int64)


def hard_answers(n, a, salt=0):
    u = unit_uniform(n, name_seed("answers", salt))
    return np.floor(u * a).astype(np.int64)


def soft_answers(n, a, salt=0):
    """(n,a) fp32 rows summing to 1 (sparse, like VQA soft scores)."""
    u = unit_uniform(n * a, name_seed("soft_answers", salt)).reshape(n, a)
    s = np.where(u > 0.97, u, 0.0)
    s[:, 0] += 1e-3
    s = s / s.sum(1, keepdims=True)
    return s.astype(np.float32)


def keep_mask(shape, p_drop, seed_name, salt=0):
    """uint8 keep-mask (1 = keep) with P(drop)=p_drop, for train-mode parity."""
    n = int(np.prod(shape))
    u = unit_uniform(n, name_seed("mask:" + seed_name, salt))
    return (u >= p_drop).astype(np.uint8).reshape(shape)
