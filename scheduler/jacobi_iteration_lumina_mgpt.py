"""Re-export of sjd_amd.scheduler.jacobi_iteration_lumina_mgpt (reference import path)."""
from sjd_amd.scheduler.jacobi_iteration_lumina_mgpt import *  # noqa: F401,F403
from sjd_amd.scheduler import jacobi_iteration_lumina_mgpt as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
