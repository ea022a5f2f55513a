"""See shim/README.md.  ``patch_reference()`` rebinds the reference's Warp-based ``get_pnts_in_grids`` (nerf/utils.py:355-443) to the HIP one."""


def patch_reference():
    import importlib
    from pienerf_amd.nerf.utils import get_pnts_in_grids
    done = []
    for name in ("nerf.utils", "nerf.renderer"):
        try:
            mod = importlib.import_module(name)
        except Exception:  # noqa: BLE001 — the reference module may not be importable in this environment
            continue
        if hasattr(mod, "get_pnts_in_grids"):
            mod.get_pnts_in_grids = get_pnts_in_grids
            done.append(name)
    return done
