"""TEST-ONLY: import the REFERENCE's own ``src/gaussian_renderer/__init__.py`` (read-only, from /root/reference) with its
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`` line
(``src/gaussian_renderer/__init__.py:15``) resolved to THIS repo's drop-in package -- the adoption path INTEGRATION.md A
describes.  ``scene.*`` is stubbed (the reference's scene package pulls in dataset readers / plyfile / simple_knn; its
render functions only use the model classes as type annotations), ``utils.*`` are the reference's own modules.
Only available where /root/reference exists (the build container); callers skip otherwise."""
import importlib
import importlib.util
import os
import sys
import types

REF = "/root/reference/src"


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "gaussian_renderer", "__init__.py"))


def load_reference_renderer():
    import gaussianhaircut_amd.diff_gaussian_rasterization as dgr
    from gaussianhaircut_amd.scene.gaussian_model import GaussianModel
    saved = {k: sys.modules.get(k) for k in ("diff_gaussian_rasterization", "scene", "scene.gaussian_model",
                                             "scene.gaussian_model_latent_strands", "utils", "utils.sh_utils",
                                             "utils.general_utils")}
    sys.modules["diff_gaussian_rasterization"] = dgr
    scene = types.ModuleType("scene")
    gm = types.ModuleType("scene.gaussian_model")
    gm.GaussianModel = GaussianModel
    # the module the reference's line 17 (`from scene.gaussian_model_latent_strands import GaussianModelHair`) resolves to
    # is this package's own compat module, not a stub: the name it asks for must exist there
    import gaussianhaircut_amd.scene.gaussian_model_latent_strands as ls
    sys.modules["scene"], sys.modules["scene.gaussian_model"] = scene, gm
    sys.modules["scene.gaussian_model_latent_strands"] = ls
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    sys.path.insert(0, REF)
    try:
        importlib.import_module("utils.sh_utils")
        importlib.import_module("utils.general_utils")
        spec = importlib.util.spec_from_file_location("ref_gaussian_renderer",
                                                      os.path.join(REF, "gaussian_renderer", "__init__.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            if saved.get(m) is None:
                sys.modules.pop(m, None)
    return mod
