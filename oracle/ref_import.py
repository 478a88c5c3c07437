"""Import the *real* reference models on CPU -- build container only.

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box, so
nothing on a `-m gpu` / smoke / bench path may import this module; it is used by
oracle/gen_golden.py (fixture generation) and by CPU tests that are skipped when
the reference is absent.

Recipe (prose): stub the third-party modules the models
import but do not need on this path (omegaconf, mcubes, wandb, torch_scatter),
make ``torch.utils.cpp_extension.load`` hand back the C oracle instead of
JIT-building the CUDA extension (app/utils/base/functions.py:14-31), and load
the model files by path so ``app/fine/__init__.py`` (hydra, imageio, trimesh,
cv2, lpips) never runs.
"""
import importlib
import importlib.util
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("ESR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "app", "fine", "model"))


def _segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    # torch_scatter.segment_coo with reduce="sum" on a sorted index == index_add
    assert reduce == "sum" and out is not None
    return out.index_add_(0, index, src)


_state = {}


def load():
    """Returns a namespace with VoxurfF, ESRNeRF, VoxurfC, base/pbr modules."""
    if _state:
        return _state["ns"]
    if not available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    from oracle import native

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    om = types.ModuleType("omegaconf")
    om.DictConfig = type("DictConfig", (dict,), {})
    om.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", om)
    sys.modules.setdefault("mcubes", types.ModuleType("mcubes"))
    wb = types.ModuleType("wandb")
    wb.config = {"system": {"debug": True, "tqdm_iters": 10}}
    sys.modules.setdefault("wandb", wb)
    ts = types.ModuleType("torch_scatter")
    ts.segment_coo = _segment_coo
    sys.modules.setdefault("torch_scatter", ts)

    import torch.utils.cpp_extension as cpp_ext

    real_load, real_name = cpp_ext.load, torch.cuda.get_device_name

    def fake_load(name, *a, **kw):
        if name == "render_utils_cuda":
            return native.as_render_utils_module()
        if name == "total_variation_cuda":
            return native.as_total_variation_module()
        raise RuntimeError(f"unexpected extension {name}")

    cpp_ext.load = fake_load
    torch.cuda.get_device_name = lambda *_a, **_k: "cpu-oracle"
    cwd = os.getcwd()
    try:
        os.chdir("/tmp")  # functions.py makedirs a build dir next to itself; keep the reference tree untouched
        _real_makedirs = os.makedirs
        os.makedirs = lambda *a, **k: None
        try:
            functions = importlib.import_module("app.utils.base.functions")
        finally:
            os.makedirs = _real_makedirs
        module = importlib.import_module("app.utils.base.module")
        pbr_module = importlib.import_module("app.utils.pbr.module")
        pbr_functions = importlib.import_module("app.utils.pbr.functions")
        image = importlib.import_module("utils2.image")
        optimizer = importlib.import_module("app.utils.optimizer")

        def by_path(modname, rel):
            spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[modname] = mod
            spec.loader.exec_module(mod)
            return mod

        voxurff = by_path("_ref_voxurff", "app/fine/model/voxurff.py")
        esrnerf = by_path("_ref_esrnerf", "app/fine/model/esrnerf.py")
        voxurfc = by_path("_ref_voxurfc", "app/coarse/model/voxurfc.py")
    finally:
        os.chdir(cwd)
        cpp_ext.load = real_load
        torch.cuda.get_device_name = real_name

    ns = types.SimpleNamespace(
        VoxurfF=voxurff.VoxurfF, ESRNeRF=esrnerf.ESRNeRF, VoxurfC=voxurfc.VoxurfC,
        functions=functions, module=module, pbr_module=pbr_module,
        pbr_functions=pbr_functions, image=image, optimizer=optimizer,
    )
    _state["ns"] = ns
    return ns
