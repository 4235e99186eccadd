"""Mirror of moge/model/__init__.py:9-18 of the reference."""
import importlib
from typing import Literal, Type


def import_model_class_by_version(version: Literal["v1", "v2"] = "v2") -> Type:
    assert version in ("v1", "v2"), f"Unsupported model version: {version}"
    return importlib.import_module(f".{version}", __package__).MoGeModel
