from pienerf_amd.gridencoder import GridEncoder  # noqa: F401  (gridencoder/__init__.py of the reference)
