"""CPU oracle for the early-exit Conformer encoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``early_exit_transformer_amd/`` may
import this package.  The only legal importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` --
and there only as the checker / the timed CPU baseline, never as the thing
shipped.  The product path has no CPU fallback: it raises when the HIP
library is missing.

Parity status: the reference-owned half of the path (subsampling, positional
encoding, length->mask, exit loop, heads, cat) is PINNED by importing the
reference's own ``Early_conformer`` / ``full_conformer`` class bodies in the
build container (tests/test_oracle_vs_reference.py, tests/golden/make_golden.py).
The third-party half (torchaudio ``Conformer``, absent from the reference tree
and version-unpinned) is restated from torchaudio's published module tree out
of the very ``torch.nn`` primitives it is made of; the reference holds no
golden vectors for it.
"""
