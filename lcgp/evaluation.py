"""`from lcgp import evaluation` / `from lcgp.evaluation import rmse, ...` (reference module path)."""
from lcgp_amd.evaluation import rmse, normalized_rmse, dss, intervalstats  # noqa: F401
