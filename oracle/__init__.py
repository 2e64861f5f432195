"""Test infrastructure: CPU restatement of the reference arithmetic (see oracle/xfm_oracle.py). Not product code."""
