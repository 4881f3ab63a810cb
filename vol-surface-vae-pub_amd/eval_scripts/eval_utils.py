"""Import-path mirror of the reference's eval_scripts/eval_utils.py (:8, :42) on the HIP path."""
from vsv.drivers import mc_generate_surfaces as generate_surfaces  # noqa: F401
from vsv.drivers import mc_generate_surfaces_multiday as generate_surfaces_multiday  # noqa: F401
