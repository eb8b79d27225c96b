"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/kaamer_oracle.c)."""
