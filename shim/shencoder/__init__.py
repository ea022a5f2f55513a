from pienerf_amd.shencoder import SHEncoder  # noqa: F401  (shencoder/__init__.py of the reference)
