from .restoration import RestorationConfig, TVDConfig, VolumeAveragingConfig  # noqa: F401
