"""CPU oracle package (test infrastructure only — see oracle/hevc_oracle.h)."""
