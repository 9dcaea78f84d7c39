"""CPU oracle for the SpeechSplit hot path (TEST INFRASTRUCTURE ONLY).

Nothing in ``speechsplit_amd/`` imports this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported baseline, never as the thing
shipped or measured as the product.

Parity status: PINNED.  ``oracle/gen_fixtures.py`` imported the reference
(``/root/reference``, this container only) and wrote ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against them.
"""
