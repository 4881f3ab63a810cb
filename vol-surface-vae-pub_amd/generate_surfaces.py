"""Import-path mirror of the reference's generate_surfaces.py (functions :13 and :60) on the HIP path."""
from vsv.drivers import generate_surfaces_multiday, generate_surfaces_random  # noqa: F401
