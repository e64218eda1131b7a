"""Model registry with the reference's contract (``sparse_caption/models/__init__.py:13-55``): classes announce themselves
with ``@register_model("<name>")`` and callers obtain them with ``get_model("<name>")``; an unknown name is a ``ValueError``
that lists what exists, a second registration of a name is refused."""
MODEL_REGISTRY = {}


def register_model(name):
    key = name.lower()

    def decorator(cls):
        if key in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model: `{name}`.")
        MODEL_REGISTRY[key] = cls
        return cls

    return decorator


def get_model(name: str):
    cls = MODEL_REGISTRY.get(name.lower())
    if cls is None:
        raise ValueError(f"Model specified `{name.lower()}` is invalid. Available options are: \n" + "\n".join(MODEL_REGISTRY))
    return cls


def register_into(reference_models_module, suffix="_hip"):
    """Drop-in hook: add these classes to the REFERENCE's registry (``sparse_caption.models``) under
    ``relation_transformer_hip`` / ``relation_transformer_prune_hip`` / ``transformer_hip``.  Their
    ``COLLATE_FN`` is this package's collate class with the reference's constructor signature
    (``COLLATE_FN(config=..., tokenizer=..., cache_dict=...)``, utils/training.py:78-81), so the reference's training module
    builds its data loaders from them unchanged."""
    for name, cls in MODEL_REGISTRY.items():
        reference_models_module.MODEL_REGISTRY.setdefault(name + suffix, cls)


from . import relation_transformer, relation_transformer_prune, transformer  # noqa: E402,F401
