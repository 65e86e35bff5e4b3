"""CPU oracle for the KMC hot path -- TEST INFRASTRUCTURE ONLY (see cet_oracle.c)."""
