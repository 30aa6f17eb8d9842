"""`from lcgp.lcgp import LCGP` (reference module path)."""
from lcgp_amd.lcgp import LCGP  # noqa: F401
