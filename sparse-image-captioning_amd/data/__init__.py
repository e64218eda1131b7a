from .collate import UpDownCollate, ObjectRelationCollate, ListDataset  # noqa: F401
