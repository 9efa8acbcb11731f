from .boardfinder import BoardFinder, GobanCorners  # noqa: F401
