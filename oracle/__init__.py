"""CPU oracle (test infrastructure).  See oracle/sdr_oracle.cpp for its scope and pinning."""
