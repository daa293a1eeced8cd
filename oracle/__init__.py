"""CPU parity oracle -- test infrastructure only (see oracle/sdirt_oracle.c)."""
