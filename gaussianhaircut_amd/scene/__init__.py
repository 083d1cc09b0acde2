from .gaussian_model import GaussianModel  # noqa: F401
from .cameras import Camera, make_camera, ring_cameras  # noqa: F401
