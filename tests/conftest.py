import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the tests marked slow (minutes of GPU time: regeneration of the committed PSNR statistics)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="slow: needs --runslow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of GPU time; skipped unless --runslow is given")
    # the CPU oracle is the checker of the GPU tests; a GPU box hands one GPU's share of its host cores (16) to this
    # process while os.cpu_count() reports all of them -- torch's default thread count oversubscribes 8x there
    if torch.cuda.device_count() > 0:
        torch.set_num_threads(min(16, os.cpu_count() or 16))


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_params():
    z = load_npz("fine_g16_params.npz")
    meta = {k: z.pop(k) for k in list(z) if k.startswith("__")}
    return {k: torch.from_numpy(v) for k, v in z.items()}, meta


@pytest.fixture(scope="session", params=["fine_g16_axis", "fine_g16_oblique", "fine_g16_prune_axis", "fine_g16_prune_oblique",
                                        "fine_g16_prune_oblique_nobg", "fine_g16_prune_oblique_gradalpha"])
def golden_case(request):
    z = load_npz(request.param + ".npz")
    return request.param, {k: torch.from_numpy(np.asarray(v)) for k, v in z.items()}


def golden_scene(name, z):
    """The slab scene a fine_g16_* fixture was generated on (oracle/gen_golden.py CASES)."""
    from esr_nerf_amd.synthetic import slab_scene
    return slab_scene("g16", oblique="oblique" in name, s_val=float(z["in/s_val"]),
                      mask="prune" if "_prune" in name else "full")


def golden_alpha_mode(name):
    """cfg neus_alpha of a fine_g16_* fixture."""
    return "grad" if name.endswith("_gradalpha") else "interp"


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """SURVEY.md 8(d) parity metric: max|a-b| / max(|b|) (rel-to-max-norm)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if b.numel() == 0:
        return 0.0 if a.numel() == 0 else float("inf")
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
