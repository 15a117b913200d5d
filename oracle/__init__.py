"""CPU oracle for the VQ-VAE train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and there only as the checker / the timed CPU baseline.  The
product path (``speech-masters-thesis_amd/``) never imports this package and
fails loudly when its HIP library is missing.

Every function restates one piece of the reference algorithm
(vliu15/speech-masters-thesis) and cites the reference file:line it follows.
The restatement is pinned by the golden vectors under ``tests/golden/`` which
were produced by importing the reference itself on CPU
(``tests/golden/make_golden.py``).
"""
