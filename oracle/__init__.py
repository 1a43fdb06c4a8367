"""CPU oracle — test infrastructure only (see oracle/svpc_oracle.py header)."""
