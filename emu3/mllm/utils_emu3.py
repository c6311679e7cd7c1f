"""reference import path emu3.mllm.utils_emu3: Emu3PrefixConstrainedLogitsHelper (reference emu3/mllm/utils_emu3.py:19-62).  Here
it only carries the grammar's constants; renew_solver turns it into the kernel-rule descriptor EOLLogitProcessor3d."""
from sjd_amd.scheduler.jacobi_iteration_emu3 import Emu3PrefixConstrainedLogitsHelper  # noqa: F401
