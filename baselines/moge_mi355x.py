"""Evaluation plugin for the reference's benchmark harness: `baselines/moge.py` with the MI355X-native model behind it.

`moge/scripts/eval_baseline.py` / `infer_baseline.py` load a standalone file by path
(`import_file_as_module`, moge/utils/tools.py:285), call `Baseline.load.main(args, standalone_mode=False)`
(eval_baseline.py:40-42) and then `baseline.infer_for_evaluation(image[, intrinsics])` per sample (eval_baseline.py:65-71).
This file makes that harness drive `moge_amd`'s `MoGeModel.infer()` unchanged:

    python moge/scripts/eval_baseline.py --baseline baselines/moge_mi355x.py --config configs/eval/all_benchmarks.json \
        --output eval_output/moge_mi355x.json --pretrained Ruicheng/moge-2-vitl-normal --version v2 --fp16

It mirrors `baselines/moge.py` option for option and quirk for quirk (a drop-in must not "fix" the plugin it replaces):
  * options and defaults (baselines/moge.py:29-37): `--num_tokens` (None), `--resolution_level` (9), `--pretrained` ("Ruicheng/moge-vitl"),
    `--fp16`, `--device` ("cuda:0"), `--version` {v1, v2}, DEFAULT v1 -> `import_model_class_by_version(version)` (:17-18);
  * the model is never `.half()`-ed: `--fp16` only becomes `use_fp16=` of `infer_for_evaluation`'s call, i.e. autocast on fp32 weights
    (:69; `MOGE_FP16` mode here).  `moge infer --fp16` is the caller that halves the model (scripts/infer.py:84), not this plugin;
  * `infer()` passes neither `use_fp16` nor `resolution_level` (:47): it runs with the model's defaults (use_fp16=True, resolution_level=9),
    whatever the flags say; `infer_for_evaluation()` passes `use_fp16` and still not `resolution_level` (:69) - the option is accepted and unused;
  * result keys (:49-60, :71-82): v1 -> `points_scale_invariant`, `depth_scale_invariant`, `intrinsics`; v2 -> `points_metric`, `depth_metric`,
    `intrinsics`; nothing else (no mask).
The ONE deliberate deviation: `utils3d.pt.intrinsics_to_fov` (baselines/moge.py:44; un-vendored dependency, absent here) is restated for
normalised intrinsics as fov_x = 2 atan(0.5 / fx)."""
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
                 device: str = "cuda:0", version: str = "v1"):
        super().__init__()
        from moge_amd.model import import_model_class_by_version
        MoGeModel = import_model_class_by_version(version)
        self.version = version
        self.model = MoGeModel.from_pretrained(pretrained_model_name_or_path).to(device).eval()
        self.device = torch.device(device)
        self.num_tokens = num_tokens
        self.resolution_level = resolution_level
        self.use_fp16 = use_fp16

    @click.command()
    @click.option("--num_tokens", type=int, default=None)
    @click.option("--resolution_level", type=int, default=9)
    @click.option("--pretrained", "pretrained_model_name_or_path", type=str, default="Ruicheng/moge-vitl")
    @click.option("--fp16", "use_fp16", is_flag=True)
    @click.option("--device", type=str, default="cuda:0")
    @click.option("--version", type=str, default="v1")
    @staticmethod
    def load(num_tokens: Optional[int], resolution_level: int, pretrained_model_name_or_path: str, use_fp16: bool, device: str = "cuda:0",
             version: str = "v1"):
        return Baseline(num_tokens, resolution_level, pretrained_model_name_or_path, use_fp16, device, version)

    def _keys(self, out: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.version == "v1":
            return {"points_scale_invariant": out["points"], "depth_scale_invariant": out["depth"], "intrinsics": out["intrinsics"]}
        return {"points_metric": out["points"], "depth_metric": out["depth"], "intrinsics": out["intrinsics"]}

    @torch.inference_mode()
    def infer(self, image: torch.Tensor, intrinsics: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        fov_x = None if intrinsics is None else _fov_x_degrees(intrinsics)
        return self._keys(self.model.infer(image, fov_x=fov_x, apply_mask=True, num_tokens=self.num_tokens))

    @torch.inference_mode()
    def infer_for_evaluation(self, image: torch.Tensor, intrinsics: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        fov_x = None if intrinsics is None else _fov_x_degrees(intrinsics)
        return self._keys(self.model.infer(image, fov_x=fov_x, apply_mask=False, num_tokens=self.num_tokens, use_fp16=self.use_fp16))
