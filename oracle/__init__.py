"""CPU oracle for the audio -> frame-event-logits hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.

PARITY UNPINNED: the reference (kasper0406/audio-to-midi) ships no tests, golden
vectors or fixtures for this path, and it cannot be executed in the build
container (jax / equinox / optax and the Rust ``modelutil`` plugin are absent).
The oracle is therefore a restatement of the reference's algorithm from its
source text, held together by

  * two independent restatements that must agree (``np_model`` numpy float64,
    written as loops/einsums that follow ``model.py`` line by line, and
    ``torch_model`` built from stock ``torch.nn.functional`` ops), and
  * hand-derivable known-answer tests (``tests/test_oracle_kat.py``).

Every function cites the reference file:line it follows.
"""
