"""Drop-in import path of the reference's lumina_mgpt package (inference solver only)."""
