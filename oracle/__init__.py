"""CPU oracles (test infrastructure only; see the module docstrings)."""
