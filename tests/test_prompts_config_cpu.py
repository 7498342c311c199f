"""CPU: the host mirrors of the reference's prompt producers and of its configuration object against the
golden vectors made by running the REFERENCE functions (src/adapters/utils.py:37-71,117-160, res_srdiff.py:125-130) on the
stub tokenizer / text encoder (tests/golden/make_golden.py::gen_prompts)."""
import hashlib
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mri-diffusion-superresolution_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

G = np.load(os.path.join(ROOT, "tests", "golden", "prompt_embeds.npz"))


def _batch():
    b = json.loads(str(G["batch_json"]))
    txt = [np.array(c) if i in b["ndarray_items"] else c for i, c in enumerate(b["txt"])]
    return {"txt": txt}


class _Accel:
    device = torch.device("cpu")


def test_fixed_prompt_embeds_match_reference():
    from mrisr import prompts
    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer
    tok, enc = StubTokenizer(), StubTextEncoder(dim=768, seed=501)
    e = prompts.get_fixed_prompt_embeds(tok, enc, _Accel())
    assert tok.seen == [["medical mri scan, high resolution"]] and enc.calls == 1
    assert e.shape == (1, 77, 768) and np.array_equal(e.numpy(), G["fixed"])
    assert prompts.get_fixed_prompt_embeds(tok, enc, _Accel(), prompt="x").shape == (1, 77, 768) and tok.seen[-1] == ["x"]


@pytest.mark.parametrize("tag", ["train", "eval", "dropall"])
def test_compute_embeddings_sd1x5_matches_reference(tag):
    """Same captions chosen (dropout + choice), same tensor, and the global `random` stream left at the same position."""
    from mrisr import prompts
    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer
    a = json.loads(str(G[f"{tag}_args"]))
    tok, enc = StubTokenizer(), StubTextEncoder(dim=768, seed=501)
    random.seed(a["seed"])
    out = prompts.compute_embeddings_sd1x5(_batch(), a["proportion_empty_prompts"], [enc], [tok], torch.device("cpu"), is_train=a["is_train"])
    assert set(out) == {"prompt_embeds"}
    pe = out["prompt_embeds"]
    assert tok.seen[-1] == json.loads(str(G[f"{tag}_captions"]))
    assert random.random() == float(G[f"{tag}_next_random"])
    assert tuple(pe.shape) == (6, 77, 768) and pe.dtype == torch.float32
    assert np.array_equal(pe[:, ::4, ::24].numpy(), G[f"{tag}_embeds_small"])
    assert hashlib.sha256(pe.numpy().tobytes()).hexdigest() == str(G[f"{tag}_embeds_sha256"])
    if tag == "dropall":
        assert tok.seen[-1] == [""] * 6
    if tag == "eval":
        assert tok.seen[-1][1] == "axial T1w" and tok.seen[-1][2] == "low field 64mT"  # caption[0] when not training


def test_encode_prompt_drops_entries_of_other_types_like_the_reference():
    from mrisr import prompts
    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer
    tok, enc = StubTokenizer(), StubTextEncoder(dim=32, seed=1)
    random.seed(0)
    e = prompts.encode_prompt_sd1x5(["a", 7, None, ["b", "c"]], [enc], [tok], 0.0, is_train=False)
    assert tok.seen[-1] == ["a", "b"] and e.shape == (2, 77, 32)


def test_train_config_defaults_are_the_notebooks_cell_and_log_configs_matches_reference():
    from mrisr.config import TrainConfig, log_configs
    c11 = json.loads(str(G["c11_config_json"]))
    cfg = TrainConfig()
    import dataclasses
    assert {f.name for f in dataclasses.fields(cfg)} == set(c11)
    for k, v in c11.items():  # (PyYAML reads the cell's `1e-5` / `1e-08` as strings: compare numerically)
        assert getattr(cfg, k) == (float(v) if isinstance(v, str) and isinstance(getattr(cfg, k), float) else v), k
    assert TrainConfig.from_dict(c11) == cfg
    logged = log_configs(cfg)
    assert list(logged) == json.loads(str(G["log_configs_keys"]))          # same keys, same order
    assert logged == json.loads(str(G["log_configs_json"]))                # = reference log_configs(TrainConfig.from_dict(c11))
    json.dumps(logged)                                                      # what a tracker needs: JSON-serialisable
    with pytest.raises(KeyError):
        TrainConfig.from_dict({"learning_rate": 1e-4, "not_a_key": 1})
    # onto the package's own objects
    cfg = TrainConfig(lora_rank=4, lora_alpha=4, mixed_precision="no")
    assert cfg.compute_dtype() == "f32" and TrainConfig().compute_dtype() == "bf16"
    assert cfg.optimizer_kwargs() == {"lr": 1e-5, "betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-8, "max_grad_norm": 1.0}
    from mrisr import DDPMScheduler
    s = DDPMScheduler(**cfg.scheduler_kwargs())
    s.set_timesteps(20)
    assert int(s.timesteps[0]) == 999 and float(s.alphas_cumprod[-1]) == 0.0  # trailing spacing, zero terminal SNR


def test_train_config_yaml_roundtrip(tmp_path):
    import yaml
    from mrisr.config import TrainConfig
    c11 = json.loads(str(G["c11_config_json"]))
    p = tmp_path / "config.xyz"
    p.write_text(yaml.safe_dump(c11))
    assert TrainConfig.from_yaml(str(p)) == TrainConfig()
