"""The training configuration object of the reference's notebook (``notebooks/ResDif_execution.ipynb`` cell 11, written to
``config.xyz`` and read back as an attribute namespace) with the SAME key names and defaults, so that the reference's
``log_configs(config)`` (``src/adapters/utils.py:37-71``) and every ``config.<key>`` access of the training cell work unchanged
on it, plus ``log_configs`` itself as a method-free mirror and the mapping onto this package's trainer / scheduler arguments."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union


@dataclass
class TrainConfig:
    # -- models / data (c11:2-12) --
    pretrained_model_name_or_path: str = "sd-legacy/stable-diffusion-v1-5"
    pretrained_vae_model_name_or_path: Optional[str] = None
    revision: Optional[str] = None
    tokenizer_name: Optional[str] = None
    output_dir: str = "/content"
    data_dir: Union[str, List[str]] = field(default_factory=lambda: ["/content/drive/MyDrive/GenAI/Data/rawdata_BIDS_3T.zip"])
    slice_axis: int = 1
    seed: int = 42
    modality: Optional[str] = None
    resolution: int = 256
    crops_coords_top_left_h: int = 0
    crops_coords_top_left_w: int = 0
    # -- loop (c11:13-24) --
    train_batch_size: int = 2
    num_train_epochs: int = 1000
    max_train_steps: int = 1000
    checkpointing_steps: int = 200
    logging_steps: int = 10
    validation_steps: int = 10
    gradient_accumulation_steps: int = 1
    gradient_checkpointing: bool = True
    learning_rate: float = 1e-5
    scale_lr: bool = False
    config: str = ""
    # -- LR schedule / optimiser (c11:25-34) --
    lr_scheduler_name: str = "cosine"
    lr_warmup_steps: int = 500
    lr_num_cycles: int = 1
    lr_power: float = 1.0
    dataloader_num_workers: int = 0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_weight_decay: float = 1e-2
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    # -- run-time switches (c11:35-43) --
    logging_dir: str = "gs://tum_genai_bucket/logs/"
    report_to: str = "wandb"
    use_8bit_adam: bool = True
    mixed_precision: str = "fp16"
    allow_tf32: bool = True
    enable_xformers_memory_efficient_attention: bool = False
    set_grads_to_none: bool = False
    proportion_empty_prompts: float = 0.1
    tracker_project_name: str = "mri_t2i_adapter_v1.5"
    # -- noise schedule (c11:44-46) --
    ddpm_scheduler_prediction_type: str = "epsilon"
    ddpm_scheduler_timestep_spacing: str = "trailing"
    ddpm_scheduler_rescale_betas_zero_snr: bool = True
    # -- LoRA (c11:47-48) --
    lora_alpha: Optional[float] = None
    lora_rank: Optional[int] = None

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "TrainConfig":
        known = {f.name for f in dataclasses.fields(cls)}
        unknown = sorted(set(d) - known)
        if unknown:
            raise KeyError(f"unknown config keys {unknown}")
        # YAML 1.1 readers (PyYAML) hand `1e-5` / `1e-08` over as strings (no dot); the notebook's reader yields floats
        numeric = {f.name: (float if f.type == "float" else int) for f in dataclasses.fields(cls) if f.type in ("float", "int")}
        d = {k: (numeric[k](v) if k in numeric and isinstance(v, str) else v) for k, v in d.items()}
        return cls(**d)

    @classmethod
    def from_yaml(cls, path: str) -> "TrainConfig":
        import yaml
        with open(path) as f:
            return cls.from_dict(yaml.safe_load(f) or {})

    # ---- onto this package's objects ----
    def scheduler_kwargs(self) -> Dict[str, Any]:
        """Arguments of ``mrisr.DDPMScheduler`` for this run (the notebook builds diffusers' DDPMScheduler from the same three keys)."""
        return {"prediction_type": self.ddpm_scheduler_prediction_type, "timestep_spacing": self.ddpm_scheduler_timestep_spacing,
                "rescale_betas_zero_snr": self.ddpm_scheduler_rescale_betas_zero_snr}

    def optimizer_kwargs(self) -> Dict[str, Any]:
        """Arguments of ``mrisr.LoRATrainer`` / ``AdapterTrainer`` (AdamW + clip, c11:29-34)."""
        return {"lr": self.learning_rate, "betas": (self.adam_beta1, self.adam_beta2), "weight_decay": self.adam_weight_decay,
                "eps": self.adam_epsilon, "max_grad_norm": self.max_grad_norm}

    def compute_dtype(self) -> str:
        """The reference's only reduced-precision hook is ``mixed_precision`` (fp16 autocast on a T4); here reduced precision is bf16
        storage with f32 accumulation, "no" is the f32 parity engine."""
        return "f32" if self.mixed_precision in ("no", None, "") else "bf16"


def log_configs(config) -> Dict[str, Any]:
    """Mirror of ``src/adapters/utils.py:37-71``: the subset of the configuration that is logged with a run, under the same keys."""
    keys = ("slice_axis", "seed", "pretrained_model_name_or_path", "pretrained_vae_model_name_or_path", "tokenizer_name", "resolution",
            "crops_coords_top_left_h", "crops_coords_top_left_w", "train_batch_size", "num_train_epochs", "max_train_steps",
            "checkpointing_steps", "gradient_accumulation_steps", "learning_rate", "scale_lr", "lr_scheduler_name", "lr_warmup_steps",
            "lr_num_cycles", "lr_power", "adam_beta1", "adam_beta2", "adam_weight_decay", "adam_epsilon", "max_grad_norm",
            "proportion_empty_prompts", "ddpm_scheduler_prediction_type", "ddpm_scheduler_timestep_spacing",
            "ddpm_scheduler_rescale_betas_zero_snr")
    out: Dict[str, Any] = {"data_dir": str(config.data_dir)}
    out.update({k: getattr(config, k) for k in keys})
    out["lora_alpha"] = getattr(config, "lora_alpha", None)
    out["lora_rank"] = getattr(config, "lora_rank", None)
    return out
