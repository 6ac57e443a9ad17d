"""The independent round-form checker lives in oracle/ (bench.py's `configs` block uses it for cfg5 too); re-exported here for
the tests that import it from this package."""
from oracle.round_form import round_form  # noqa: F401
