"""MI355X-native triangle pipeline (gfx950 HIP) behind the C API of the
reference software renderer.  See include/mi355r.h for the boundary."""
from .binding import Pipeline, View, GPU_LIB, ASSETS, SCENES, REPO  # noqa: F401
