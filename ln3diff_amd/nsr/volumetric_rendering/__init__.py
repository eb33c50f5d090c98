from .renderer import ImportanceRenderer  # noqa: F401
