"""Mirror of the reference's ``scheduler/`` package (the drop-in boundary)."""
