from pienerf_amd.raymarching import *  # noqa: F401,F403  (raymarching/__init__.py of the reference: from .raymarching import *)
