"""CPU oracle for the latent-diffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stablediffusion_amd/`` may import this
package: it is the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

What it is: a plain fp32 PyTorch *restatement* (functional, state-dict driven) of
the reference's UNet / VAE-decoder / DDIM path, following the reference file:line
cited on every function, with the minimal repairs of SURVEY.md §8(a-bugs) for the
composites that crash as shipped.

How it is pinned: ``tools/gen_golden.py`` imports the reference's own leaf modules
from ``/root/reference`` (through a loader shim, nothing is copied) and writes
seeded input/weight/output vectors to ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against them.  The
reference ships no tests or fixtures of its own, so those goldens
are the only pin: anything they do not cover (the DDIM loop body, apply_model,
decode_first_stage, PLMS - all absent from the reference) is "parity unpinned"
and restated from the upstream semantics the reference's stubs point at.
"""
