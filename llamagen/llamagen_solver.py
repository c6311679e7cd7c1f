"""Re-export of sjd_amd.llamagen_solver (reference import path llamagen.llamagen_solver)."""
from sjd_amd.llamagen_solver import *  # noqa: F401,F403
from sjd_amd.llamagen_solver import (LlamaGenSolver, renew_llamagen, generate, MaxlenCriteria, sample, top_k_top_p_filtering,  # noqa: F401
                                     logits_to_probs)
