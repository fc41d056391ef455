"""CPU oracle package -- test infrastructure only (see oracle/gato_oracle.c)."""
