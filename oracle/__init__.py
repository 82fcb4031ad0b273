"""CPU oracle (test infrastructure only) — see sift_oracle.c."""
