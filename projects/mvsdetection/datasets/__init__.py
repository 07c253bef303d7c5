from .tsdf import coordinates  # noqa: F401
