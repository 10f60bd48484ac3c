"""Oracle package root (test infrastructure only; see oracle/ctrlv_ref)."""
