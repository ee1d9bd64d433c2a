"""Parity oracle (test infrastructure). See oracle/i2sdf_oracle.py."""
