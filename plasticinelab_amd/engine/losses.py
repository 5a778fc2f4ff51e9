"""Mirror of plb.engine.losses.Loss (/root/reference/plb/engine/losses/loss.py).

The mass scatter, density / SDF / contact terms, their adjoint and the target-SDF
sweeps run on the GPU (plmpm_loss_*); this class keeps the reference's
bookkeeping (cumulative ``loss``, ``_start_loss``, ``_last_loss``, reward and
incremental IoU, loss.py:269-302) in plain Python.
"""
from __future__ import annotations

import os

import numpy as np


class Loss:
    def __init__(self, cfg, sim):
        self.cfg = cfg
        self.sim = sim
        self.engine = sim.engine
        self.res, self.n_grid, self.dx, self.dim = sim.res, sim.n_grid, sim.dx, sim.dim
        self.n_particles = sim.n_particles
        self.primitives = [p for p in sim.primitives if p.action_dim > 0]      # loss.py:20-24
        self.inf = 1000
        self.loss = 0.0            # cumulative, like the reference's `loss` field
        self.sdf_loss = self.density_loss = self.contact_loss = 0.0
        self.soft_contact_loss = False
        self._weights = (10.0, 10.0, 1.0)
        self._iou = self._target_iou = 0.0
        self._start_loss = self._last_loss = 0.0
        self._init_iou = 0.0
        self.target_density = None

    # ---- targets / weights
    def load_target_density(self, path=None, grids=None):                      # loss.py:46-57
        if path is not None and len(path) == 0 and grids is None:
            return                      # no target configured yet (set one with load_target_density(grids=...))
        if path is not None or grids is not None:
            if path is not None and len(path) > 0:
                grids = np.load(path if os.path.isabs(path) or os.path.exists(path)
                                else os.path.join(os.path.dirname(os.path.abspath(__file__)), "../", path))
            grids = np.asarray(grids, np.float64)
            if grids.shape != tuple(self.res):
                raise ValueError(f"target grid shape {grids.shape} does not match the simulator grid {self.res}")
            self.target_density = grids
            self.engine.loss_set_target(grids)                                 # update_target, loss.py:103-106
            g = grids
            ma = g.max()
            I = (g * g).sum() / ma / ma                                        # iou(target, target), loss.py:55-57
            U = 2 * g.sum() / ma
            self._target_iou = I / (U - I)

    def initialize(self):                                                      # loss.py:59-66
        w = self.cfg.weight
        self.set_weights(w.sdf, w.density, w.contact, self.cfg.soft_contact)
        self.load_target_density(self.cfg.target_path)

    def set_weights(self, sdf, density, contact, is_soft_contact):             # loss.py:68-72
        self._weights = (float(sdf), float(density), float(contact))
        self.soft_contact_loss = bool(is_soft_contact)
        self.engine.loss_set_weights(sdf, density, contact, self.soft_contact_loss)

    @property
    def target_sdf(self):
        return self.engine.target_sdf()

    # ---- kernels
    def compute_loss_kernel(self, f):                                          # loss.py:186-208
        out = self.engine.loss_forward(f)
        self.loss += out["loss"]
        self.sdf_loss, self.density_loss, self.contact_loss = out["sdf_loss"], out["density_loss"], out["contact_loss"]
        self._iou = out["iou"]
        return out

    def compute_loss_kernel_grad(self, f):                                     # loss.py:210-237
        self.engine.loss_backward(f)

    def iou(self):                                                             # loss.py:260-262
        return self._iou

    def _extract_loss(self, f):                                                # loss.py:269-279
        self.compute_loss_kernel(f)
        return {"loss": self.loss, "contact_loss": self.contact_loss, "density_loss": self.density_loss,
                "sdf_loss": self.sdf_loss, "iou": self._iou, "target_iou": self._target_iou}

    def reset(self):                                                           # loss.py:281-286
        self.clear_loss()
        info = self._extract_loss(0)
        self._start_loss = info["loss"]
        self._init_iou = info["iou"]
        self._last_loss = 0

    def compute_loss(self, f):                                                 # loss.py:288-298
        info = self._extract_loss(f)
        r = self._start_loss - (info["loss"] - self._last_loss)
        cur_step_loss = info["loss"] - self._last_loss
        self._last_loss = info["loss"]
        denom = info["target_iou"] - self._init_iou
        inc = max(min((info["iou"] - self._init_iou) / denom, 1), 0) if denom != 0 else 0.0
        info["reward"] = r
        info["incremental_iou"] = inc
        info["loss"] = cur_step_loss
        return info

    def clear_loss(self):                                                      # loss.py:182-184
        self.loss = 0.0

    def clear(self):                                                           # loss.py:300-302
        self.clear_loss()
        self._last_loss = 0
