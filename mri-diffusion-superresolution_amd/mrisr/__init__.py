"""mrisr - MI355X-native denoiser hot path for MRI diffusion super-resolution (host-side Python over libmrisr.so).

Importing the package does not need a GPU; constructing a model or calling an op does, and raises
``MrisrError`` when the HIP library or device is missing.  There is no CPU fallback anywhere in this package."""
from ._lib import LIB_PATH, MrisrError  # noqa: F401
from .models import Adapter_XL, ControlNetModel, UNet2DConditionModel, UNetConfig  # noqa: F401
from .pipeline import (Sampler, decode_to_vis, get_res_shifting_latents, log_validation,  # noqa: F401
                       prepare_condition_image)
from .schedulers import DDIMScheduler, DDPMScheduler  # noqa: F401
from .train import LoRATrainer, cosine_lr  # noqa: F401
from .vae import AutoencoderKL, VAEConfig, vae_param_shapes  # noqa: F401
from .metrics import MRIEvaluator  # noqa: F401
from .train import AdapterTrainer, ControlNetTrainer, joint_step, joint_step_overlapped  # noqa: F401
from .dist import BucketedReducer  # noqa: F401
from .datasets import (FastMRILazyDataset, SliceDataset, gaussian_blur, get_data_dicts_artificial,  # noqa: F401
                       pad_or_center_crop, resize_slices, simulate_low_field)
from .prompts import compute_embeddings_sd1x5, encode_prompt_sd1x5, get_fixed_prompt_embeds  # noqa: F401
from .config import TrainConfig, log_configs  # noqa: F401
