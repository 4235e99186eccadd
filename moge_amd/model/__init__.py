"""Mirror of moge/model/__init__.py:9-18 of the reference."""
import importlib
from typing import Literal, Type


def import_model_class_by_version(version: Literal["v1", "v2"] = "v2") -> Type:
    if version == "v1":
        raise NotImplementedError("moge_amd implements the MoGe-2 (v2) inference path only")
    if version != "v2":
        raise ValueError(f'Unsupported model version: {version}')
    return importlib.import_module(".v2", __package__).MoGeModel
