"""Deterministic stand-ins for the CLIP tokenizer / text encoder (TEST INFRASTRUCTURE).

The reference's prompt producers (``src/adapters/utils.py:117-160``, ``src/adapters/res_srdiff.py:125-130``) call a
``transformers`` CLIP tokenizer and ``CLIPTextModel`` whose weights need the network.  Their CONTROL FLOW - caption dropout,
the choice among alternative captions, padding to ``model_max_length``, ``[0]`` of the encoder output, device moves - is what the
product has to mirror; these stubs give it something deterministic to run on: the tokenizer hashes characters into ids, the
encoder is a seeded embedding table plus a position term.  Used by tests/golden/make_golden.py (driving the REFERENCE functions)
and by the tests (driving the product's mirrors of them)."""
from __future__ import annotations

import torch


class _BatchEncoding(dict):
    """What ``tokenizer(...)`` returns: attribute access + ``.to(device)`` (res_srdiff.py:126-127 calls ``inputs.to(...)``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return _BatchEncoding({k: v.to(device) for k, v in self.items()})


class StubTokenizer:
    model_max_length = 77
    vocab_size = 4096

    def __init__(self):
        self.seen = []  # every caption list it was called with (lets a test read back the dropout / choice decisions)

    def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
        caps = [text] if isinstance(text, str) else list(text)
        self.seen.append(list(caps))
        assert padding == "max_length" and truncation and return_tensors == "pt", "the reference's call convention"
        L = max_length or self.model_max_length
        ids = torch.zeros((len(caps), L), dtype=torch.int64)
        for b, c in enumerate(caps):
            toks = [1] + [2 + (ord(ch) * 131 + 7 * i) % (self.vocab_size - 3) for i, ch in enumerate(c)][: L - 2] + [self.vocab_size - 1]
            ids[b, : len(toks)] = torch.tensor(toks)
        return _BatchEncoding(input_ids=ids, attention_mask=(ids != 0).to(torch.int64))


class StubTextEncoder:
    """``text_encoder(input_ids)[0]`` -> ``[B, 77, dim]`` (CLIPTextModel's last_hidden_state position)."""

    def __init__(self, dim: int = 768, seed: int = 501, dtype=torch.float32, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.table = torch.randn((StubTokenizer.vocab_size, dim), generator=g).to(device=device, dtype=dtype)
        self.pos = (0.1 * torch.randn((StubTokenizer.model_max_length, dim), generator=g)).to(device=device, dtype=dtype)
        self.device = torch.device(device)
        self.calls = 0

    def __call__(self, input_ids, **kw):
        self.calls += 1
        assert input_ids.device.type == self.device.type, "ids must have been moved to the encoder's device"
        h = self.table[input_ids] + self.pos[None, : input_ids.shape[1]]
        return (h, h[:, -1])
