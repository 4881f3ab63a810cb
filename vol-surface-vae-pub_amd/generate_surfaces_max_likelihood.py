"""Import-path mirror of the reference's generate_surfaces_max_likelihood.py (z = 0 generation, :13-103);
the batched device passes live in vsv/drivers.py."""
from vsv.drivers import generate_surfaces_mle, generate_surfaces_multiday_mle  # noqa: F401
