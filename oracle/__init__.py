"""CPU oracle (test infrastructure only — see oracle/gencore_oracle.h).  Never imported by gencore_amd/."""
