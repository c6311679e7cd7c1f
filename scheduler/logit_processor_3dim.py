"""Re-export of sjd_amd.scheduler.logit_processor_3dim (reference import path)."""
from sjd_amd.scheduler.logit_processor_3dim import *  # noqa: F401,F403
from sjd_amd.scheduler import logit_processor_3dim as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
