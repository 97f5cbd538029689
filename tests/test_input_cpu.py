"""Oracle of the input-pipeline tail (ToTensor + Normalize + flips) against a plain numpy float32 restatement."""
import numpy as np
import torch

from oracle.input_ref import IMAGENET_MEAN, IMAGENET_STD, to_tensor_normalize


def test_oracle_matches_numpy_float32_arithmetic():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(3, 10, 12, 3), dtype=np.uint8)
    got = to_tensor_normalize(torch.from_numpy(x)).numpy()
    v = x.astype(np.float32) / np.float32(255)
    m = np.asarray(IMAGENET_MEAN, np.float32)
    s = np.asarray(IMAGENET_STD, np.float32)
    want = ((v - m) / s).transpose(0, 3, 1, 2)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # every uint8 level of every channel
    lv = np.tile(np.arange(256, dtype=np.uint8)[None, None, :, None], (1, 1, 1, 3))
    got = to_tensor_normalize(torch.from_numpy(lv)).numpy()
    assert np.array_equal(got, ((lv.astype(np.float32) / np.float32(255) - m) / s).transpose(0, 3, 1, 2))


def test_oracle_flip_flags():
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.integers(0, 256, size=(4, 6, 8, 3), dtype=np.uint8))
    base = to_tensor_normalize(x)
    got = to_tensor_normalize(x, torch.tensor([0, 1, 2, 3], dtype=torch.uint8))
    assert torch.equal(got[0], base[0])
    assert torch.equal(got[1], base[1].flip(-1))
    assert torch.equal(got[2], base[2].flip(-2))
    assert torch.equal(got[3], base[3].flip(-1).flip(-2))
