"""The ``[restoration]`` section of a workflow configuration (reference:
presets/workflows/config/restoration.py:12-117, helpers of config/utils.py:47-70): which restoration runs and with which
options.  Same fields, defaults, conversions and errors as the reference; TOML is read with ``tomllib`` (``tomli`` before
Python 3.11)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Union

try:  # Python >= 3.11
    import tomllib
except ModuleNotFoundError:  # pragma: no cover - depends on the interpreter
    import tomli as tomllib


def _deep_merge(dst: dict, src: dict) -> dict:
    for key, value in src.items():
        if isinstance(value, dict) and isinstance(dst.get(key), dict):
            _deep_merge(dst[key], value)
        else:
            dst[key] = value
    return dst


def _section_from_toml(path, section: str) -> dict:
    if isinstance(path, Path):
        data = tomllib.loads(path.read_text())
    elif isinstance(path, list):
        data = {}
        for p in path:
            _deep_merge(data, tomllib.loads(p.read_text()))
    else:
        raise TypeError(f"Path must be a Path or list of Paths. It is {type(path)}.")
    if section not in data:
        raise KeyError(f"Section '{section}' not found in the configuration.")
    return data[section]


def _key(section: dict, key: str, default=None, required: bool = True, type_=None):
    if required and key not in section:
        raise KeyError(f"Missing key '{key}' in section {section}.")
    if key in section:
        return type_(section[key]) if type_ else section[key]
    return default


@dataclass
class VolumeAveragingConfig:
    rev_size: int = 3

    def load(self, sec: dict) -> "VolumeAveragingConfig":
        self.rev_size = _key(sec, "rev_size", self.rev_size, required=False)
        return self


@dataclass
class TVDConfig:
    """Options of the TVD restoration: ``method`` (one of the four ``TVD`` methods), ``weight`` (a number, or
    "image_porosity" / "boolean_porosity" to take the rig's porosity as a heterogeneous weight), ``max_num_iter``,
    ``eps``, and for "heterogeneous bregman" ``omega`` and ``regularization``; unknown keys are kept in ``kwargs`` and
    forwarded to the operator."""

    method: str = "chambolle"
    weight: Union[float, str] = 0.1
    max_num_iter: int = 200
    eps: float = 2e-4
    omega: float = 1.0
    regularization: float = 1.0
    kwargs: dict = field(default_factory=dict)

    def load(self, sec: dict) -> "TVDConfig":
        self.method = _key(sec, "method", self.method, required=False, type_=str)
        raw = _key(sec, "weight", self.weight, required=False)
        self.weight = raw if isinstance(raw, str) else float(raw)
        self.max_num_iter = _key(sec, "max_num_iter", self.max_num_iter, required=False, type_=int)
        self.eps = _key(sec, "eps", self.eps, required=False, type_=float)
        self.omega = _key(sec, "omega", self.omega, required=False, type_=float)
        self.regularization = _key(sec, "regularization", self.regularization, required=False, type_=float)
        known = {f.name for f in dataclasses.fields(self)} - {"kwargs"}
        self.kwargs = {k: v for k, v in sec.items() if k not in known}
        return self


@dataclass
class RestorationConfig:
    method: Optional[str] = "volume_average"
    options: Union[VolumeAveragingConfig, TVDConfig, None] = None
    ignore: list = field(default_factory=list)

    def load(self, path) -> "RestorationConfig":
        sec = path if isinstance(path, dict) else _section_from_toml(path, "restoration")
        self.method = _key(sec, "method", required=True, type_=str).lower()
        if self.method == "none":
            self.method = None
        if self.method not in ("volume_average", "tvd", None):
            raise NotImplementedError(f"Invalid restoration method: {self.method}")
        self.ignore = _key(sec, "ignore", default=[], required=False, type_=list)
        if not all(isinstance(entry, str) for entry in self.ignore):
            raise ValueError("restoration.ignore must be a list of strings.")
        options = sec.get("options", {})
        if self.method == "volume_average":
            self.options = VolumeAveragingConfig().load(options)
        elif self.method == "tvd":
            self.options = TVDConfig().load(options)
        return self
