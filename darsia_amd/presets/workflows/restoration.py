"""Restoration objects from a workflow configuration (reference: presets/workflows/restoration.py:21-157).

The TVD branch builds a ``darsia_amd.TVD``; with a porosity weight or an ignore mask the weight becomes an array and the
method "heterogeneous bregman" -- the array-coefficient device pipeline.  Unlike the reference, whose ``TVD`` forwards
``regularization`` twice and fails, the operator built here runs.  Volume averaging is not part of the
TV path: that branch assembles its inputs exactly like the reference and hands them to ``VolumeAveraging`` / ``REV`` of
the namespace given as ``backend`` (e.g. the ``darsia`` module)."""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np

import darsia_amd

from .config.restoration import RestorationConfig, TVDConfig, VolumeAveragingConfig

logger = logging.getLogger(__name__)


class RestorationMaskFactory:
    """Predefined ignore masks of a rig: pixels the restoration must not smooth across."""

    def __init__(self, fluidflower) -> None:
        self.fluidflower = fluidflower
        self._ignore_mask_builders = {
            "image_porosity": lambda: self.fluidflower.image_porosity.img <= 0,
            "boolean_porosity": lambda: ~self.fluidflower.boolean_porosity.img.astype(bool),
            "inner_labels": lambda: ~self.fluidflower.inner_labels.img.astype(bool),
        }

    def build_ignore_mask(self, mask_names: list) -> Optional[np.ndarray]:
        ignore_mask = None
        for name in mask_names:
            if name not in self._ignore_mask_builders:
                raise ValueError(f"Unknown restoration ignore mask '{name}'. "
                                 f"Valid values are: {list(self._ignore_mask_builders.keys())}.")
            current = self._ignore_mask_builders[name]()
            ignore_mask = current if ignore_mask is None else np.logical_or(ignore_mask, current)
        return ignore_mask


def build_restoration(restoration_config: Optional[RestorationConfig], fluidflower, backend=None):
    """A callable restoration object, or None if none is configured.  ``fluidflower`` provides ``baseline``,
    ``image_porosity``, ``boolean_porosity`` (and ``inner_labels`` for that ignore mask)."""
    if restoration_config is None:
        logger.info("No restoration configuration found. Proceeding without restoration.")
        return None
    method = restoration_config.method
    if method is None:
        return None
    backend = darsia_amd if backend is None else backend
    ignore_mask = RestorationMaskFactory(fluidflower).build_ignore_mask(restoration_config.ignore)
    active_mask = None if ignore_mask is None else (~ignore_mask).astype(float)

    if method == "volume_average":
        options = restoration_config.options
        if not isinstance(options, VolumeAveragingConfig):
            options = VolumeAveragingConfig()
        averaging, rev = getattr(backend, "VolumeAveraging", None), getattr(backend, "REV", None)
        if averaging is None or rev is None:
            raise NotImplementedError("volume averaging is outside the TV path of darsia_amd: pass backend=darsia")
        image_porosity = fluidflower.image_porosity.copy()   # the rig shares its porosity image
        if active_mask is not None:
            image_porosity.img *= active_mask
        return averaging(rev=rev(size=options.rev_size, img=fluidflower.baseline), mask=image_porosity)

    if method == "tvd":
        options = restoration_config.options
        if not isinstance(options, TVDConfig):
            options = TVDConfig()
        tvd_method = options.method
        if isinstance(options.weight, str):
            if options.weight == "image_porosity":
                weight = fluidflower.image_porosity
            elif options.weight == "boolean_porosity":
                weight = fluidflower.boolean_porosity
            else:
                raise ValueError(f"Unknown weight string '{options.weight}'. "
                                 "Valid string values are 'image_porosity' and 'boolean_porosity'. "
                                 "For a scalar weight, provide a numeric value in the config.")
            tvd_method = "heterogeneous bregman"
        else:
            weight = options.weight
        if hasattr(weight, "img"):
            weight = weight.img
        if active_mask is not None:
            weight = np.multiply(weight, active_mask)
            tvd_method = "heterogeneous bregman"
        return backend.TVD(method=tvd_method, weight=weight, max_num_iter=options.max_num_iter, eps=options.eps,
                           omega=options.omega, regularization=options.regularization, **options.kwargs)

    raise NotImplementedError(f"Restoration method '{method}' not supported.")
