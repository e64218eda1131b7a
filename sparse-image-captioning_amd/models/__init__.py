"""Model registry — same contract as the reference's ``sparse_caption/models/__init__.py:13-55``:
``register_model(name)`` decorator, ``get_model(name)`` with a ``ValueError`` listing the options."""
MODEL_REGISTRY = {}


def register_model(name):
    def register_model_cls(cls):
        if name in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model: `{name}`.")
        MODEL_REGISTRY[name.lower()] = cls
        return cls

    return register_model_cls


def get_model(name: str):
    name = name.lower()
    try:
        return MODEL_REGISTRY[name]
    except KeyError:
        _list = "\n".join(MODEL_REGISTRY.keys())
        raise ValueError(f"Model specified `{name}` is invalid. Available options are: \n{_list}")


def register_into(reference_models_module, suffix="_hip"):
    """Drop-in hook: add these classes to the REFERENCE's registry (``sparse_caption.models``) under
    ``relation_transformer_hip`` / ``relation_transformer_prune_hip``."""
    for name, cls in MODEL_REGISTRY.items():
        key = name + suffix
        if key not in reference_models_module.MODEL_REGISTRY:
            reference_models_module.MODEL_REGISTRY[key] = cls


from . import relation_transformer, relation_transformer_prune, transformer  # noqa: E402,F401
