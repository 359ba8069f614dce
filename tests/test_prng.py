"""The integer PRNG must be bit-stable: golden capture (authoring container), oracle and GPU box all
regenerate weights/inputs from it."""
import numpy as np
import torch

from wacv23_tsnet_amd import prng


def test_hash_known_answers():
    h = prng._hash64_numpy(3, prng.fnv1a64("img_enc.model.1.weight"), 1000, lane=2)
    assert int(h[0]) == 0x554C28A2F97705E8 and int(h[999]) == 0xC7D57694F223B759
    assert prng.fnv1a64("") == 0xCBF29CE484222325


def test_torch_and_numpy_hash_agree():
    for seed, name, lane in [(0, "a", 0), (7, "dec.map_conv.weight", 1), (123456789, "x" * 50, 2)]:
        t = prng.hash64(seed, prng.fnv1a64(name), 4097, lane=lane).numpy().view(np.uint64)
        n = prng._hash64_numpy(seed, prng.fnv1a64(name), 4097, lane=lane)
        assert np.array_equal(t, n)


def test_distributions_and_known_values():
    w = prng.normal(0, "a.weight", (64, 8, 7, 7))
    assert w.dtype == torch.float32 and abs(w.std().item() - 0.02) < 5e-4 and abs(w.mean().item()) < 5e-4
    assert w.flatten()[:3].tolist() == [0.009070740081369877, -0.011789245530962944, -0.023276062682271004]
    u = prng.uniform01(1, "x", (10000,))
    assert 0.0 <= u.min().item() and u.max().item() < 1.0 and abs(u.mean().item() - 0.5) < 0.02
    b = prng.bernoulli(1, "b", (10000,))
    assert set(b.unique().tolist()) == {0.0, 1.0} and abs(b.mean().item() - 0.5) < 0.02
    # streams are independent of shape factorisation and of each other
    assert torch.equal(prng.uniform01(1, "x", (100, 100)).flatten(), u)
    assert not torch.equal(prng.uniform01(2, "x", (10000,)), u)
