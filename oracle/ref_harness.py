"""Loads the UNMODIFIED reference (/root/reference/fsw_embedding.py, fsw_conv.py) on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: /root/reference does not exist on
the GPU box, so everything importing this module must skip when `available()` is False.

Two third-party imports of the reference are absent from this image and cannot be installed:
  * type_enforced  (fsw_embedding.py:85)  -- used solely as `@type_enforced.Enforcer(enabled=True)`
    runtime argument-type checking on a handful of methods; no arithmetic.
  * torch_geometric (fsw_conv.py:4-9)     -- FSW_conv derives from MessagePassing but never calls
    propagate(); message/aggregate/update are empty stubs (fsw_conv.py:374-381).
As SURVEY.md section 8(c) prescribes, in-memory placeholder modules (a decorator that returns its
argument; an nn.Module base class) are registered in sys.modules before the reference sources are
executed by importlib.  No reference file is copied, edited or written; bytecode caching is disabled.
The custom CUDA library is not loaded (load_custom_cuda_lib=False / module flag cleared), so
segcumsum takes the reference's pure-torch branch (fsw_embedding.py:2847-2850).
"""
import importlib.util
import os
import sys
import types

REF_DIR = "/root/reference"


def available():
    return os.path.isfile(os.path.join(REF_DIR, "fsw_embedding.py"))


def _install_placeholders():
    import torch

    if "type_enforced" not in sys.modules:
        te = types.ModuleType("type_enforced")

        class Enforcer:  # decorator factory: @type_enforced.Enforcer(enabled=True)
            def __init__(self, enabled=True):
                pass

            def __call__(self, obj):
                return obj

        te.Enforcer = Enforcer
        sys.modules["type_enforced"] = te

    if "torch_geometric" not in sys.modules:
        pyg = types.ModuleType("torch_geometric")
        nn_mod = types.ModuleType("torch_geometric.nn")
        utils = types.ModuleType("torch_geometric.utils")
        gg = types.ModuleType("torch_geometric.graphgym")
        reg = types.ModuleType("torch_geometric.graphgym.register")

        class MessagePassing(torch.nn.Module):
            def __init__(self, aggr=None, **kw):
                super().__init__()

        def _identity_decorator(name):
            return lambda cls: cls

        nn_mod.MessagePassing = MessagePassing
        utils.add_self_loops = None
        utils.degree = None
        gg.cfg = None
        gg.register = reg
        reg.register_layer = _identity_decorator
        reg.register_pooling = _identity_decorator
        pyg.nn, pyg.utils, pyg.graphgym = nn_mod, utils, gg
        sys.modules.update({
            "torch_geometric": pyg, "torch_geometric.nn": nn_mod, "torch_geometric.utils": utils,
            "torch_geometric.graphgym": gg, "torch_geometric.graphgym.register": reg})


_cache = {}


def load():
    """Returns (fsw_embedding module, fsw_conv module) of the reference, CPU-only configuration."""
    if "mods" in _cache:
        return _cache["mods"]
    if not available():
        raise RuntimeError("reference not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    _install_placeholders()
    spec = importlib.util.spec_from_file_location("ref_fsw_conv", os.path.join(REF_DIR, "fsw_conv.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)          # fsw_conv.py:28-30 loads fsw_embedding.py itself
    emb = conv.fsw_embedding
    # FSW_conv does not forward load_custom_cuda_lib (fsw_conv.py:314-323): clear the module flag so
    # that the NVIDIA-only libfsw_embedding.so is never dlopen'ed (prebuilt binary: never loaded).
    emb.fsw_embedding_produce_error_on_custom_library_loading_failure = False
    emb.libfsw_embedding_path = "/nonexistent/libfsw_embedding.so"
    _cache["mods"] = (emb, conv)
    return emb, conv
