"""seaborn stand-in (TEST INFRASTRUCTURE ONLY)."""
