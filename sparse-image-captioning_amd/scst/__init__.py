from .scorers import CaptionScorer  # noqa: F401
