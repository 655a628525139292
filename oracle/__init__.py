"""CPU oracle of the hot path -- TEST INFRASTRUCTURE ONLY (see subspace_oracle.py header).  PARITY UNPINNED."""
