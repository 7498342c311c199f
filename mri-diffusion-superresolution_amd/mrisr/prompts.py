"""Host mirror of the reference's ``encoder_hidden_states`` producers - same names, arguments, RNG
consumption and return contracts:

  get_fixed_prompt_embeds   src/adapters/res_srdiff.py:125-130   one prompt -> ``[1, 77, 768]`` on ``accelerator.device``
  encode_prompt_sd1x5       src/adapters/utils.py:117-145        caption dropout / choice -> tokenizer -> ``text_encoder(ids)[0]``
  compute_embeddings_sd1x5  src/adapters/utils.py:149-160        ``{"prompt_embeds": [B, 77, 768] on device}``

The tokenizer and the CLIP text encoder are the caller's objects (CLIP is outside the hot path): this module is
the glue between them and the denoiser, kept here so that ``src/adapters`` call sites switch over by changing the import only.
The product consumes what these return as they return it - any float dtype, any device, ``fixed_embeds[0:1]`` slices -
``UNet2DConditionModel.forward`` / ``Sampler.run`` / ``LoRATrainer`` move and convert on the way in (tests/test_gpu_prompts.py)."""
from __future__ import annotations

import random
from typing import Dict, Sequence

import numpy as np
import torch

FIXED_PROMPT = "medical mri scan, high resolution"  # res_srdiff.py:125


def _tokenize(tokenizer, text):
    return tokenizer(text, padding="max_length", max_length=tokenizer.model_max_length, truncation=True, return_tensors="pt")


@torch.no_grad()
def get_fixed_prompt_embeds(tokenizer, text_encoder, accelerator, prompt: str = FIXED_PROMPT) -> torch.Tensor:
    """The embedding ``log_validation`` is handed as ``fixed_embeds`` (it slices ``[0:1]``)."""
    ids = _tokenize(tokenizer, prompt).to(accelerator.device).input_ids
    return text_encoder(ids)[0]


def _pick_captions(prompt_batch: Sequence, proportion_empty_prompts: float, is_train: bool):
    """One ``random.random()`` per caption first (dropout), and only for a kept list/array caption one ``random.choice`` when
    training - the global ``random`` stream is consumed exactly as utils.py:119-126 consumes it; entries of any other type
    are dropped from the batch, as there."""
    out = []
    for cap in prompt_batch:
        if random.random() < proportion_empty_prompts:
            out.append("")  # classifier-free-guidance dropout
        elif isinstance(cap, str):
            out.append(cap)
        elif isinstance(cap, (list, np.ndarray)):
            out.append(random.choice(cap) if is_train else cap[0])
    return out


@torch.no_grad()
def encode_prompt_sd1x5(prompt_batch, text_encoders, tokenizers, proportion_empty_prompts, is_train: bool = True) -> torch.Tensor:
    captions = _pick_captions(prompt_batch, proportion_empty_prompts, is_train)
    tokenizer, text_encoder = tokenizers[0], text_encoders[0]  # SD-1.5: the one and only pair
    ids = _tokenize(tokenizer, captions).input_ids
    return text_encoder(ids.to(text_encoder.device))[0]


def compute_embeddings_sd1x5(batch, proportion_empty_prompts, text_encoders, tokenizers, device, is_train: bool = True) -> Dict[str, torch.Tensor]:
    emb = encode_prompt_sd1x5(batch["txt"], text_encoders, tokenizers, proportion_empty_prompts, is_train)
    return {"prompt_embeds": emb.to(device)}
