"""CPU oracle (test infrastructure only; never imported by xgcm_amd)."""
