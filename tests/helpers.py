"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import os

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def sample_idx(numel, tag, n=2048):
    """Same seeded sampling as tests/golden/make_golden.py."""
    g = np.random.default_rng([0x5A3917, tag])
    return g.integers(0, numel, size=min(n, numel))


def load_cfg(yaml_name):
    with open(os.path.join(REPO, "models", "transformer", yaml_name)) as f:
        return yaml.safe_load(f)


def state_dict_shapes_from_oracle_cfg(cfg):
    """Build the reference-compatible state_dict layout from our own (CPU-constructible) Model."""
    from icafusion_amd.models.yolo import Model
    return Model(cfg)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
