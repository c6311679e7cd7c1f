"""Re-export of sjd_amd.scheduler.jacobi_iteration_anhole (reference import path)."""
from sjd_amd.scheduler.jacobi_iteration_anhole import *  # noqa: F401,F403
from sjd_amd.scheduler import jacobi_iteration_anhole as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
