"""Re-export of sjd_amd.inference_solver (reference import path lumina_mgpt.inference_solver)."""
from sjd_amd.inference_solver import FlexARInferenceSolver  # noqa: F401
