"""`from lcgp.covmat import Matern32` (reference module path)."""
from lcgp_amd.covmat import Matern32  # noqa: F401
