"""Drop-in import path of the reference (`from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler`)."""
