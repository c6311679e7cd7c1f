"""Drop-in import path of the reference's llamagen package (solver + GPT registry only)."""
