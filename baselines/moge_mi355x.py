"""Evaluation plugin for the reference's benchmark harness.

`moge/scripts/eval_baseline.py` / `infer_baseline.py` load a standalone file by path
(`import_file_as_module`, moge/utils/tools.py:285), call `Baseline.load.main(args, standalone_mode=False)`
(eval_baseline.py:40-42) and then `baseline.infer_for_evaluation(image[, intrinsics])` per sample (eval_baseline.py:65-71).
This file makes that harness drive the MI355X-native `MoGeModel.infer()` unchanged:

    python moge/scripts/eval_baseline.py --baseline baselines/moge_mi355x.py --config configs/eval/all_benchmarks.json \
        --output eval_output/moge_mi355x.json --pretrained Ruicheng/moge-2-vitl-normal --fp16

Contract mirrored from moge/test/baseline.py:7-42 (MGEBaselineInterface): `.device`, a click-decorated static `load`,
`infer(image, intrinsics=None) -> dict`, `infer_for_evaluation(...)`; v2 models report `points_metric`, `depth_metric`,
`intrinsics` (baselines/moge.py:54-59).  The un-vendored `utils3d.pt.intrinsics_to_fov` (baselines/moge.py:44) is restated for
normalised intrinsics: fov_x = 2 atan(0.5 / fx)."""
from typing import Dict, Optional

import click
import torch

try:                                              # inside the reference checkout: be a real subclass
    from moge.test.baseline import MGEBaselineInterface as _Base
except Exception:                                 # stand-alone (this repo): same duck type
    class _Base:                                  # noqa: D401
        device: torch.device


def _fov_x_degrees(intrinsics: torch.Tensor) -> torch.Tensor:
    fx = intrinsics[..., 0, 0].float()
    return torch.rad2deg(2.0 * torch.atan(0.5 / fx))


class Baseline(_Base):
    def __init__(self, num_tokens: Optional[int], resolution_level: int, pretrained_model_name_or_path: str, use_fp16: bool,
                 device: str = "cuda:0"):
        super().__init__()
        from moge_amd.model import import_model_class_by_version
        MoGeModel = import_model_class_by_version("v2")
        self.model = MoGeModel.from_pretrained(pretrained_model_name_or_path).to(device).eval()
        if use_fp16:
            self.model.half()                     # what `moge infer --fp16` does (moge/scripts/infer.py:84)
        self.device = torch.device(device)
        self.num_tokens = num_tokens
        self.resolution_level = resolution_level
        self.use_fp16 = use_fp16

    @click.command()
    @click.option("--num_tokens", type=int, default=None)
    @click.option("--resolution_level", type=int, default=9)
    @click.option("--pretrained", "pretrained_model_name_or_path", type=str, default="Ruicheng/moge-2-vitl-normal")
    @click.option("--fp16", "use_fp16", is_flag=True)
    @click.option("--device", type=str, default="cuda:0")
    @staticmethod
    def load(num_tokens: Optional[int], resolution_level: int, pretrained_model_name_or_path: str, use_fp16: bool, device: str = "cuda:0"):
        return Baseline(num_tokens, resolution_level, pretrained_model_name_or_path, use_fp16, device)

    def _run(self, image: torch.Tensor, intrinsics: Optional[torch.Tensor], apply_mask: bool) -> Dict[str, torch.Tensor]:
        fov_x = None if intrinsics is None else _fov_x_degrees(intrinsics)
        out = self.model.infer(image, fov_x=fov_x, apply_mask=apply_mask, num_tokens=self.num_tokens,
                               resolution_level=self.resolution_level, use_fp16=self.use_fp16)
        res = {"points_metric": out["points"], "depth_metric": out["depth"], "intrinsics": out["intrinsics"]}
        if "mask" in out:
            res["mask"] = out["mask"]
        return res

    @torch.inference_mode()
    def infer(self, image: torch.Tensor, intrinsics: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        return self._run(image, intrinsics, apply_mask=True)

    @torch.inference_mode()
    def infer_for_evaluation(self, image: torch.Tensor, intrinsics: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        return self._run(image, intrinsics, apply_mask=False)
