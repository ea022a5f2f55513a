"""Headless training loop on the HIP ops — what ``main_train.py`` + ``Trainer.train_step / train_one_epoch`` do for
the ``-O`` (cuda_ray) configuration, without the dataset / GUI / tensorboard baggage.

Reference: main_train.py:60-80 (Adam(get_params(lr), betas=(0.9, 0.99), eps=1e-15), LambdaLR 0.1^(iter/iters), MSE), nerf/trainer.py:158-207
(train_step: render(..., perturb=True, force_all_rays=False), per-ray MSE), :604-667 (train_one_epoch: update_extra_state every
``update_extra_interval`` steps, zero_grad / backward / step / scheduler per step).  Data comes as posed images already on the device
(there is no dataset on the box): ``RayImageSet`` samples ``num_rays`` random pixels of a random view per step like
nerf/provider.py's training collate (one view per batch, rays drawn uniformly, :270-300).
"""
import torch

from .nerf.utils import get_rays


class RayImageSet:
    """Posed RGB(A) images [V, H, W, C] resident on the device; ``batch(num_rays)`` -> rays + target colours of one random view."""

    def __init__(self, poses, intrinsics, images, generator=None):
        self.poses = poses.to(torch.float32)                # [V, 4, 4] cam2world
        self.intrinsics = intrinsics                        # (fx, fy, cx, cy)
        self.images = images.to(torch.float32)              # [V, H, W, 3 or 4]
        self.V, self.H, self.W = images.shape[:3]
        self.gen = generator

    def batch(self, num_rays):
        dev = self.images.device
        v = int(torch.randint(0, self.V, (1,), generator=self.gen, device="cpu"))
        rays = get_rays(self.poses[v:v + 1], self.intrinsics, self.H, self.W)          # the full view, then a random subset of pixels
        inds = torch.randint(0, self.H * self.W, (num_rays,), generator=self.gen, device="cpu").to(dev)
        return {"rays_o": rays["rays_o"][:, inds], "rays_d": rays["rays_d"][:, inds], "images": self.images[v].reshape(1, -1, self.images.shape[-1])[:, inds],
                "index": v}


class ParamEMA:
    """Exponential moving average of the trainable parameters with torch_ema's interface subset the reference uses (trainer.py:86-89,
    643-644, 751-753, 789-790: update / store / copy_to / restore / state_dict), including torch_ema's warm-up of the decay,
    min(decay, (1 + n) / (10 + n))."""

    def __init__(self, parameters, decay=0.95):
        self.decay, self.num_updates = float(decay), 0
        self.params = [p for p in parameters if p.requires_grad]
        self.shadow = [p.detach().clone() for p in self.params]
        self.stored = None

    @torch.no_grad()
    def update(self):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        for s, p in zip(self.shadow, self.params):
            s.sub_((1.0 - d) * (s - p))

    @torch.no_grad()
    def store(self):
        self.stored = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def copy_to(self):
        for s, p in zip(self.shadow, self.params):
            p.copy_(s)

    @torch.no_grad()
    def restore(self):
        for s, p in zip(self.stored, self.params):
            p.copy_(s)
        self.stored = None

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": [s.clone() for s in self.shadow], "collected_params": None}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = float(sd["decay"]), int(sd["num_updates"])
        for s, v in zip(self.shadow, sd["shadow_params"]):
            s.copy_(v.to(s.device))


class Trainer:
    def __init__(self, model, opt, lr=1e-2, iters=30000, update_extra_interval=16, num_rays=4096, ema_decay=None, fp16=False):
        """ema_decay: main_train.py:78 passes 0.95; None (default) trains without an average, like Trainer's own default (trainer.py:19).
        fp16 (trainer.py:20,84: ``--fp16``): the steps run under autocast with a GradScaler — half hash tables, half nn.Linear, and the half
        scatter-add of the grid's backward (gridencoder.cu:324-331)."""
        self.model, self.opt = model, dict(opt)
        self.fp16 = bool(fp16)
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.fp16)   # trainer.py:84
        self.optimizer = torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda it: 0.1 ** min(it / iters, 1))
        self.criterion = torch.nn.MSELoss(reduction="none")
        self.update_extra_interval, self.num_rays = update_extra_interval, num_rays
        self.global_step = 0
        self.ema = ParamEMA(model.parameters(), ema_decay) if ema_decay is not None else None

    def _render_opts(self):
        keep = ("dt_gamma", "max_steps", "T_thresh")
        return {k: self.opt[k] for k in keep if k in self.opt}

    def train_step(self, data):
        """trainer.py:158-207 (the image-supervised branch)."""
        images = data["images"]
        C = images.shape[-1]
        if C == 4:  # random per-pixel background under the alpha matte (trainer.py:193-196)
            bg_color = torch.rand_like(images[..., :3])
            gt_rgb = images[..., :3] * images[..., 3:] + bg_color * (1 - images[..., 3:])
            bg = bg_color.view(-1, 3)
        else:
            bg, gt_rgb = 1, images
        outputs = self.model.render(data["rays_o"], data["rays_d"], staged=False, bg_color=bg, perturb=True, force_all_rays=False, **self._render_opts())
        pred_rgb = outputs["image"]
        loss = self.criterion(pred_rgb, gt_rgb).mean(-1).mean()
        return pred_rgb, gt_rgb, loss

    def train(self, dataset, steps):
        """``steps`` iterations of trainer.py:625-645; returns the per-step losses."""
        self.model.train()
        losses = []
        for _ in range(steps):
            if self.model.cuda_ray and self.global_step % self.update_extra_interval == 0:
                with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):   # trainer.py:629
                    self.model.update_extra_state()
            self.global_step += 1
            self.optimizer.zero_grad()
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):       # trainer.py:637
                _, _, loss = self.train_step(dataset.batch(self.num_rays))
            self.scaler.scale(loss).backward()                                         # trainer.py:640-642
            self.scaler.step(self.optimizer)
            self.scaler.update()
            self.lr_scheduler.step()
            if self.ema is not None:  # trainer.py:643-644
                self.ema.update()
            losses.append(float(loss.detach()))
        return losses

    @torch.no_grad()
    def evaluate(self, dataset, view):
        """Full-image PSNR of one view (eval() mode: the inference loop of run_cuda)."""
        self.model.eval()
        if self.ema is not None:  # evaluation runs on the averaged weights (trainer.py:751-753, 789-790)
            self.ema.store()
            self.ema.copy_to()
        rays = get_rays(dataset.poses[view:view + 1], dataset.intrinsics, dataset.H, dataset.W)
        out = self.model.render(rays["rays_o"], rays["rays_d"], bg_color=1, perturb=False, **self._render_opts())
        if self.ema is not None:
            self.ema.restore()
        img = dataset.images[view]
        gt = img[..., :3] * img[..., 3:] + (1 - img[..., 3:]) if img.shape[-1] == 4 else img
        mse = torch.mean((out["image"].view(dataset.H, dataset.W, 3) - gt) ** 2)
        return float(-10 * torch.log10(mse)), out
