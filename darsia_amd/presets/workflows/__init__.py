from .restoration import RestorationMaskFactory, build_restoration  # noqa: F401
