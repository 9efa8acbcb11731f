from .stonesfinder import StonesFinder, PosGrid  # noqa: F401
