"""Asset IO for the headless front end: checkpoints in the reference's ``.pth`` layout, PNG frames, camera poses.

Reference: nerf/trainer.py:793-854 (``save_checkpoint``: {'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'model':
state_dict[, 'optimizer', 'lr_scheduler']}), :856-916 (``load_checkpoint``: bare state dict or the dict above, ``strict=False``),
main_render.py:12-26 (``save_image``), :28-45 (``get_pose`` from transforms*.json), nerf/provider.py:19-27 (``nerf_matrix_to_ngp``).
PLY point clouds: pienerf_amd/scene.py (``read_ply`` / ``write_ply``), ``Simulator.InitializeFromPly`` / ``OutputToPly``.
"""
import glob
import json
import os

import numpy as np
import torch


def save_checkpoint(model, path, epoch=0, global_step=0, stats=None, optimizer=None, lr_scheduler=None, full=False, ema=None, scaler=None):
    """trainer.py:793-830 (the ``best=False`` branch): one ``.pth`` the reference's ``Trainer.load_checkpoint`` can read.  ``ema`` (an object
    with state_dict(), training.ParamEMA) and ``scaler`` go under the reference's 'ema' / 'scaler' keys of a full checkpoint (:806-812)."""
    state = {"epoch": int(epoch), "global_step": int(global_step), "stats": stats if stats is not None else {"checkpoints": [], "results": []}}
    if getattr(model, "cuda_ray", False):
        state["mean_count"] = model.mean_count
        state["mean_density"] = model.mean_density
    if full:
        if optimizer is not None:
            state["optimizer"] = optimizer.state_dict()
        if lr_scheduler is not None:
            state["lr_scheduler"] = lr_scheduler.state_dict()
        if scaler is not None:
            state["scaler"] = scaler.state_dict()
        if ema is not None:
            state["ema"] = ema.state_dict()
    state["model"] = model.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
    return path


def latest_checkpoint(ckpt_dir, name="ngp"):
    """trainer.py:858-860: the newest ``{name}_ep*.pth`` of a workspace's checkpoint directory, or None."""
    found = sorted(glob.glob(os.path.join(ckpt_dir, f"{name}_ep*.pth")))
    return found[-1] if found else None


def _safe_load(path, map_location, allow_pickle):
    """torch.load restricted to tensors + plain python / numpy scalars (the reference's checkpoint layout holds nothing else).  A file that
    needs arbitrary pickle globals is refused unless the caller opts in: a downloaded .pth must not be able to run code."""
    try:
        import numpy._core.multiarray as _ma  # numpy >= 2
    except ImportError:  # numpy 1.x
        import numpy.core.multiarray as _ma
    safe = [np.dtype, np.ndarray, type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64)), type(np.dtype(np.int32))]
    for name in ("_reconstruct", "scalar"):
        fn = getattr(_ma, name, None)
        if fn is not None:
            safe.append(fn)
    import pickle
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        # only what the restricted unpickler itself refuses (torch raises pickle.UnpicklingError, older versions a RuntimeError that names
        # weights_only); a missing file, a corrupt archive or a device-mapping error propagates as it is
        if isinstance(e, RuntimeError) and "weights_only" not in str(e) and "Unsupported" not in str(e):
            raise
        if not allow_pickle:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}). If you trust the file, pass allow_pickle=True "
                               "(main_render: --trust-ckpt)") from e
        import warnings
        warnings.warn(f"{path}: needs pickle globals outside the allow-list; loading with full pickle because allow_pickle=True", stacklevel=3)
        return torch.load(path, map_location=map_location, weights_only=False)


def load_checkpoint(model, path, model_only=True, optimizer=None, lr_scheduler=None, map_location=None, ema=None, allow_pickle=False, scaler=None):
    """trainer.py:856-916.  Returns dict(missing_keys, unexpected_keys, epoch, global_step).  The model is left in eval() mode."""
    ck = _safe_load(path, map_location or next(model.parameters()).device, allow_pickle)
    info = dict(missing_keys=[], unexpected_keys=[], epoch=None, global_step=None)
    if "model" not in ck:  # a bare state dict
        model.load_state_dict(ck)
    else:
        res = model.load_state_dict(ck["model"], strict=False)
        info["missing_keys"], info["unexpected_keys"] = list(res.missing_keys), list(res.unexpected_keys)
        if getattr(model, "cuda_ray", False):
            if "mean_count" in ck:
                model.mean_count = ck["mean_count"]
            if "mean_density" in ck:
                model.mean_density = ck["mean_density"]
        if not model_only:
            info["epoch"], info["global_step"] = ck.get("epoch"), ck.get("global_step")
            if optimizer is not None and "optimizer" in ck:
                optimizer.load_state_dict(ck["optimizer"])
            if lr_scheduler is not None and "lr_scheduler" in ck:
                lr_scheduler.load_state_dict(ck["lr_scheduler"])
            if scaler is not None and "scaler" in ck:  # trainer.py:911-916
                scaler.load_state_dict(ck["scaler"])
        if ema is not None and "ema" in ck:  # trainer.py:884-885 (loaded whenever present)
            ema.load_state_dict(ck["ema"])
    if hasattr(model, "_net_sig"):
        model._net_sig = None  # the packed weight image of the fused kernel is rebuilt on next use
    model.eval()
    return info


def save_image(image, path, W, H):
    """main_render.py:12-26: float RGB in [0,1] (any shape with H*W*3 elements) -> 8-bit PNG."""
    from PIL import Image
    if torch.is_tensor(image):
        image = image.detach().cpu().numpy()
    data = (np.clip(np.asarray(image, np.float32), 0, 1) * 255).astype(np.uint8).reshape(H, W, 3)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(data, "RGB").save(path)
    return data


def linear_to_srgb(x):
    """nerf/utils.py:44-46 (applied to the prediction when opt.color_space == 'linear', trainer.py:583-584)."""
    return torch.where(x < 0.0031308, 12.92 * x, 1.055 * x ** 0.41666 - 0.055)


def srgb_to_linear(x):
    """nerf/utils.py:49-51."""
    return torch.where(x < 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0, 0, 0)):
    """nerf/provider.py:19-27: blender / colmap cam2world -> the renderer's axis convention, translation scaled and offset."""
    pose = np.asarray(pose, np.float32)
    return np.array([[pose[1, 0], -pose[1, 1], -pose[1, 2], pose[1, 3] * scale + offset[0]],
                     [pose[2, 0], -pose[2, 1], -pose[2, 2], pose[2, 3] * scale + offset[1]],
                     [pose[0, 0], -pose[0, 1], -pose[0, 2], pose[0, 3] * scale + offset[2]],
                     [0, 0, 0, 1]], dtype=np.float32)


def get_pose(data_dir, frame_str):
    """main_render.py:28-45: the transform_matrix of the frame whose file_path contains ``frame_str`` (transforms_train.json, else
    transforms.json); None when neither file or no such frame exists."""
    for name in ("transforms_train.json", "transforms.json"):
        fp = os.path.join(data_dir, name)
        if os.path.exists(fp):
            with open(fp) as f:
                data = json.load(f)
            for frame in data["frames"]:
                if frame_str in frame["file_path"]:
                    return np.array(frame["transform_matrix"], dtype=np.float32)
            return None
    return None
