"""MI355X-native Object Relation Transformer hot path (drop-in for ``sparse_caption.models``).

    from sparse_image_captioning_amd.models import get_model
    model = get_model("relation_transformer")(config).cuda()

Compute runs in ``libortk.so`` (hand-written HIP for gfx950, C-ABI in ``include/ortk.h``); this package is the
host-side mirror of the reference's plugin interface (``sparse_caption/models/__init__.py:13-55``).
"""
from . import _lib  # noqa: F401
from .models import get_model, register_model, MODEL_REGISTRY  # noqa: F401
from .utils.config import Config  # noqa: F401

__version__ = "0.1.0"
