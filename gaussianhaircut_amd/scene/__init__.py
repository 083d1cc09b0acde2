from .gaussian_model import GaussianModel  # noqa: F401
from .cameras import Camera, make_camera, ring_cameras  # noqa: F401
from .gaussian_model_strands import GaussianModelCurves, GaussianModelStrands  # noqa: F401  (reference scene/__init__.py:19)
from .gaussian_model_latent_strands import GaussianModelHair  # noqa: F401  (reference scene/__init__.py:18)
