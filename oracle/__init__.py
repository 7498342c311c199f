"""CPU oracle for the MRI diffusion super-resolution hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (the ``mrisr`` package over
``libmrisr.so``) never imports anything from here and fails loudly when the
HIP library is missing.

What it restates (pure PyTorch on the CPU, fp32 or fp64):

* ``oracle.unet``       - diffusers ``UNet2DConditionModel`` / ``ControlNetModel``
                          (SD-1.5 family) + peft LoRA linears.  The arithmetic is
                          NOT in the reference repo: it lives in the un-vendored,
                          unpinned third-party ``diffusers`` (>= 0.36 by the
                          evidence in SURVEY.md App. D.2) and ``peft``.  The
                          published architecture is restated from SURVEY.md
                          App. A; call sites: reference
                          ``src/adapters/res_srdiff.py:65-78``.
* ``oracle.schedulers`` - scaled-linear beta table, leading/trailing timestep
                          spacing, DDIM(eta=0) step.
* ``oracle.sampler``    - the reference's own Res-SRDiff forward shift and
                          reverse step (``src/adapters/res_srdiff.py:7-25,84-96``).
* ``oracle.adapter``    - the reference's T2I-Adapter ``Adapter_XL`` with
                          ``sk=True`` (``src/adapters/modules.py:52-157``).

Pinning status
--------------
* sampler / forward shift / adapter: PINNED by golden vectors produced in the
  build container by importing the reference's own modules
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* UNet / ControlNet / LoRA: **parity unpinned** - no reference test, fixture or
  runnable reference implementation of diffusers exists offline.  Known-answer
  checks only: parameter counts 859,520,964 / 361,279,120 / 797,184 and the
  diffusers state-dict key/shape list (tests/test_oracle_unet.py).
"""
