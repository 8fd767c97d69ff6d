"""CPU oracle for the salt-segmentation hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) / numpy restatement of the arithmetic of the
reference's U-Net training/inference path (neptune-ai/open-solution-salt-identification).
It exists so that the hand-written HIP kernels can be checked against something that runs
everywhere, including the GPU box where /root/reference does not exist.

Rules (enforced by tests/test_host_cpu.py::test_product_never_imports_the_oracle_or_reads_the_reference):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
  * the product package never imports it and never falls back to it.

Pinning: every function here is checked against golden vectors produced by importing the
reference's own modules in the build container (tests/golden/make_golden.py, executed with
torch 2.10 CPU) and against the known-answer values KAT-1..KAT-11 recorded in SURVEY.md §8c.
Arithmetic that lives in un-vendored third-party code (torchvision 0.2.0 ResNet layout,
torch 0.3.1 operator kernels) is restated from its public definition; the reference ships no
tests of its own, so those parts are pinned only by "reference source executed on torch 2.10".

Style: pure functions over a flat ``{state_dict key: tensor}`` mapping, so the same closed-form
weights can be fed to the reference modules, to this oracle and to the HIP path by key name.
"""
from . import blocks, nets, losses, metrics, specs  # noqa: F401
