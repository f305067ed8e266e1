"""Import the unmodified reference model file in the build container (oracle tooling, own code).

Recipe of SURVEY.md section 8(c): the reference package ``models`` cannot be imported as shipped
(``src/models/__init__.py:8-17`` names six absent modules), and ``cav_mae_base.py`` needs timm/ipdb/tome,
a network download (:236) and a hard-coded weight file (:240).  So:
  1. oracle/ref_shim/ goes first on sys.path (stub timm / ipdb / tome);
  2. a synthetic package object ``models`` with __path__ = [<reference>/src/models] bypasses the broken
     __init__, plus a stub ``models.yb_tome``;
  3. ``torch.load`` returns {} for the hard-coded /mnt/... path;
  4. importlib imports ``models.cav_mae_base`` from the reference tree where it lies (nothing is copied).
Returns None when /root/reference is absent (GPU box).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("AVSIAM_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "models", "cav_mae_base.py"))


def import_reference_module():
    if not reference_available():
        return None
    if "models.cav_mae_base" in sys.modules:
        return sys.modules["models.cav_mae_base"]
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "src", "models")]
    sys.modules["models"] = pkg
    yb = types.ModuleType("models.yb_tome")
    yb.yb_bipartite_soft_matching = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["models.yb_tome"] = yb
    sys.dont_write_bytecode = True                 # never write .pyc into the read-only reference tree
    return importlib.import_module("models.cav_mae_base")


def build_reference_ft_model(label_dim):
    """Construct the reference ``CAVMAEFT_BASE(label_dim)`` on CPU (weights are loaded by the caller)."""
    return build_reference_model("CAVMAEFT_BASE", label_dim)


def build_reference_model(cls="CAVMAE_BASE", *ctor_args):
    """Construct the reference ``CAVMAE_BASE()`` on CPU (random init; weights are loaded by the caller)."""
    import torch
    mod = import_reference_module()
    if mod is None:
        return None
    real_load = torch.load

    def fake_load(path, *a, **k):
        if isinstance(path, str) and path.startswith("/mnt/"):
            return {}
        return real_load(path, *a, **k)

    torch.load = fake_load
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            model = getattr(mod, cls)(*ctor_args)
    finally:
        torch.load = real_load
    return model


class PlanRecorder:
    """Wraps the reference model's RNG-consuming methods and records the mask plan they produced, so the
    oracle/HIP path can be driven with exactly the tokens the reference kept."""

    def __init__(self, model):
        self.model = model
        self.calls = []           # (kind, ids_keep [n,keep], ids_restore [n,L])
        self.perms = []
        self._orig_u = model.random_masking_unstructured
        self._orig_s = model.random_masking_structured

    def __enter__(self):
        import torch
        rec = self

        def wrap(orig, kind):
            def f(x, *a, **k):
                xm, mask, ids_restore = orig(x, *a, **k)
                ids_shuffle = torch.argsort(ids_restore, dim=1)
                rec.calls.append((kind, ids_shuffle[:, :xm.shape[1]].clone(), ids_restore.clone()))
                return xm, mask, ids_restore
            return f

        self.model.random_masking_unstructured = wrap(self._orig_u, "u")
        self.model.random_masking_structured = wrap(self._orig_s, "s")
        self._randperm = torch.randperm

        def randperm(*a, **k):
            p = rec._randperm(*a, **k)
            rec.perms.append(p.clone())
            return p

        torch.randperm = randperm
        return self

    def __exit__(self, *exc):
        import torch
        torch.randperm = self._randperm
        del self.model.random_masking_unstructured
        del self.model.random_masking_structured
        return False
